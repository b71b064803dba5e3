"""The OpenCV fitting method on the GPU (csrc/epnp_ransac.hip through the C ABI
epos_solve_pnp_ransac / epos_solve_pnp_ransac_device; scripts/infer.py:505-528) against
its oracle (oracle/epnp_ref.c): the pose bit for bit, the inlier mask, the winning set,
its inlier count and the iteration bound."""
import ctypes

import numpy as np
import pytest

from tests.test_oracle_epnp import K, scene, _rot_err_deg

pytestmark = pytest.mark.gpu


def _both(xyz, xy, iters=400, thr=4.0, conf=0.99):
  from epos_amd import fitting
  from oracle import epnp_ref
  ok_o, P_o, mask_o, info_o = epnp_ref.solvePnPRansac(xyz, xy, K, iters, thr, conf)
  ok, rvec, tvec, inl, P, info = fitting.solvePnPRansac(
      xyz, xy, K, None, iters, thr, conf, return_pose=True, return_info=True)
  assert ok == ok_o
  assert list(info) == list(info_o)
  if ok:
    np.testing.assert_array_equal(P, P_o)                         # bit for bit
    np.testing.assert_array_equal(inl.ravel(), np.nonzero(mask_o)[0])
    np.testing.assert_allclose(fitting.Rodrigues(rvec), P[:, :3], atol=1e-9)
    np.testing.assert_array_equal(tvec, P[:, 3:])
  else:
    assert P is None and rvec is None and inl is None
  return ok, P, info


@pytest.mark.parametrize('n,outliers,sigma', [
    (6, 0.0, 0.0), (7, 0.0, 1.0), (50, 0.0, 1.0), (200, 0.3, 1.0), (500, 0.5, 1.0),
    (2000, 0.5, 1.0), (2000, 0.7, 1.0), (10000, 0.3, 1.0), (10000, 0.7, 2.0), (257, 0.2, 0.5),
    (1024, 0.9, 1.0)])
def test_same_bits_as_the_oracle(n, outliers, sigma):
  for seed in range(2):
    xyz, xy, R, t, good = scene(1000 * seed + n, n, sigma=sigma, outliers=outliers)
    ok, P, info = _both(xyz, xy)
    if outliers <= 0.5:          # 5-point sets: at 70 % outliers 400 draws often hold no clean set
      assert ok
      if n >= 50:
        assert _rot_err_deg(P[:, :3], R) < 1.0 and np.linalg.norm(P[:, 3] - t) < 0.02 * t[2]


def test_iteration_cap_and_threshold_are_honoured():
  xyz, xy, R, t, good = scene(42, 800, sigma=1.0, outliers=0.5)
  for iters, thr, conf in [(1, 4.0, 0.99), (7, 4.0, 0.99), (400, 1.0, 0.99), (400, 8.0, 0.5),
                           (1000, 4.0, 0.999999)]:
    ok, P, info = _both(xyz, xy, iters, thr, conf)
    assert info[3] <= iters


def test_degenerate_inputs():
  from epos_amd import fitting
  for n in (0, 3, 4):
    xyz, xy, R, t, _ = scene(n, n)
    ok, rvec, tvec, inl = fitting.solvePnPRansac(xyz, xy, K, None, 400, 4.0, 0.99)
    assert not ok and rvec is None and inl is None
  xyz, xy, R, t, _ = scene(5, 5)                       # exactly one minimal set
  ok, P, info = _both(xyz, xy)
  assert ok and list(info[:2]) == [0, 5]
  rng = np.random.RandomState(0)                       # clutter
  _both(rng.uniform(-50, 50, (300, 3)), rng.uniform(0, 480, (300, 2)))
  xyz, xy, R, t, _ = scene(9, 40)                      # coplanar object points: EPnP's
  xyz[:, 2] = 0.0                                      # control points degenerate
  _both(xyz, xy)
  xyz, xy, R, t, _ = scene(10, 40)                     # all correspondences identical
  _both(np.repeat(xyz[:1], 40, 0), np.repeat(xy[:1], 40, 0))
  with pytest.raises(NotImplementedError):
    fitting.solvePnPRansac(xyz, xy, K, np.array([0.1, 0, 0, 0]))
  with pytest.raises(NotImplementedError):
    fitting.solvePnPRansac(xyz, xy, K, None, flags=0)


def test_batched_device_entry_matches_per_object_calls():
  """Several objects (slots) of different sizes in ONE launch, incl. an empty slot, a slot
  below the minimal set and a slot that overflows the capacity (contained: no pose)."""
  import torch
  from epos_amd import _lib
  from oracle import epnp_ref
  lib = _lib.load()
  sizes = [300, 0, 4, 5, 1500, 77]
  scenes = [scene(50 + i, n, sigma=1.0, outliers=0.4) for i, n in enumerate(sizes)]
  xy = np.concatenate([s[1] for s in scenes] + [np.zeros((0, 2))])
  xyz = np.concatenate([s[0] for s in scenes] + [np.zeros((0, 3))])
  base = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
  S, N = len(sizes), int(base[-1])
  Ks = np.tile(K.reshape(1, 9), (S, 1))
  Ks[4] = [900.0, 0, 300.0, 0, 905.0, 250.0, 0, 0, 1]         # per-slot intrinsics
  sc4 = scenes[4]
  Y = sc4[0] @ sc4[2].T + sc4[3]
  xy4 = np.stack([900.0 * Y[:, 0] / Y[:, 2] + 300.0, 905.0 * Y[:, 1] / Y[:, 2] + 250.0], 1)
  xy[base[4]:base[5]][sc4[4]] = xy4[sc4[4]]
  d = 'cuda:0'
  p = _lib.PnpRansacParams()
  lib.epos_pnp_ransac_params_default(ctypes.byref(p))
  t_xy, t_xyz = torch.from_numpy(xy).to(d), torch.from_numpy(xyz).to(d)
  t_base, t_K = torch.from_numpy(base).to(d), torch.from_numpy(Ks).to(d)
  wb = lib.epos_pnp_ransac_workspace_bytes(S, N, ctypes.byref(p))
  assert wb > 0
  work = torch.empty(wb, dtype=torch.uint8, device=d)
  poses = torch.zeros(S, 12, dtype=torch.float64, device=d)
  succ = torch.full((S,), -7, dtype=torch.int32, device=d)
  mask = torch.full((N,), 9, dtype=torch.uint8, device=d)
  info = torch.zeros(S, 4, dtype=torch.int32, device=d)
  vp = lambda t: ctypes.c_void_p(t.data_ptr())
  st = torch.cuda.Stream()
  with torch.cuda.stream(st):
    _lib.check(lib.epos_solve_pnp_ransac_device(
        vp(t_xy), vp(t_xyz), vp(t_base), S, N, vp(t_K), ctypes.byref(p), vp(work), vp(poses),
        vp(succ), vp(mask), vp(info), ctypes.c_void_p(st.cuda_stream)), 'device entry')
  st.synchronize()
  t_poses, t_mask, t_info = poses, mask, info
  succ, poses, mask, info = succ.cpu().numpy(), poses.cpu().numpy(), mask.cpu().numpy(), info.cpu().numpy()
  for s, n in enumerate(sizes):
    ok_o, P_o, mask_o, info_o = epnp_ref.solvePnPRansac(
        xyz[base[s]:base[s + 1]], xy[base[s]:base[s + 1]], Ks[s].reshape(3, 3))
    assert bool(succ[s]) == ok_o, s
    assert list(info[s]) == list(info_o), s
    np.testing.assert_array_equal(mask[base[s]:base[s + 1]], mask_o)
    if ok_o:
      np.testing.assert_array_equal(poses[s], np.concatenate([P_o[:, :3].ravel(), P_o[:, 3]]))
  assert succ[0] == 1 and succ[4] == 1 and succ[1] == 0 and succ[2] == 0
  # capacity smaller than the pooled size: the overflowing slots are treated as empty
  succ2 = torch.full((S,), -7, dtype=torch.int32, device=d)
  _lib.check(lib.epos_solve_pnp_ransac_device(
      vp(t_xy), vp(t_xyz), vp(t_base), S, int(base[4]), vp(t_K), ctypes.byref(p), vp(work),
      vp(t_poses), vp(succ2), vp(t_mask), vp(t_info), None), 'device entry')
  torch.cuda.synchronize()
  assert list(succ2.cpu().numpy()) == [1, 0, 0, int(succ[3]), 0, 0]


def test_min_point_number_skips_small_sets():
  """EposPnpRansacParams.min_point_number: what the script's `n < 6` filter
  (infer.py:420-422) does before the cv2 call."""
  import ctypes as ct
  from epos_amd import _lib
  lib = _lib.load()
  p = _lib.PnpRansacParams()
  lib.epos_pnp_ransac_params_default(ct.byref(p))
  assert p.min_point_number == 0 and p.iterations_count == 400
  vp = ct.c_void_p
  for n, mn, want in [(5, 0, 1), (5, 6, 0), (6, 6, 1), (40, 41, 0)]:
    xyz, xy, R, t, _ = scene(n + mn, n)
    p.min_point_number = mn
    pose, mask, Kd = np.zeros(12), np.ones(n, np.uint8), np.ascontiguousarray(K).reshape(9)
    ok = lib.epos_solve_pnp_ransac(xy.ctypes.data_as(vp), xyz.ctypes.data_as(vp), n,
                                   Kd.ctypes.data_as(vp), ct.byref(p),
                                   pose.ctypes.data_as(vp), mask.ctypes.data_as(vp), None)
    assert ok == want, (n, mn)
    assert mask.sum() == (n if want else 0)


def test_independent_of_launch_history():
  xyz, xy, R, t, good = scene(77, 3000, sigma=1.0, outliers=0.5)
  first = _both(xyz, xy)[1]
  for _ in range(3):
    np.testing.assert_array_equal(_both(xyz, xy)[1], first)
