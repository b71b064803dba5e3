"""The oracle of the OpenCV fitting method (oracle/epnp_ref.c; scripts/infer.py:505-528)
against an independent numpy statement of EPnP / the RANSAC loop and known poses. cv2 is
absent from this image: the oracle is UNPINNED against OpenCV itself (see its header)."""
import numpy as np
import pytest

from oracle import epnp_ref
from tests.helpers import epnp_numpy

K = np.array([[1066.8, 0, 313.0], [0, 1067.5, 241.3], [0, 0, 1]])


def _rot(rng):
  q = rng.normal(size=4)
  q /= np.linalg.norm(q)
  w, x, y, z = q
  return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def scene(seed, n, sigma=0.0, outliers=0.0):
  rng = np.random.RandomState(seed)
  R = _rot(rng)
  t = np.array([rng.uniform(-80, 80), rng.uniform(-60, 60), rng.uniform(500, 1100)])
  xyz = rng.uniform(-1, 1, size=(n, 3)) * np.array([60.0, 45.0, 30.0])
  Y = xyz @ R.T + t
  xy = np.stack([K[0, 0] * Y[:, 0] / Y[:, 2] + K[0, 2], K[1, 1] * Y[:, 1] / Y[:, 2] + K[1, 2]], 1)
  xy += rng.normal(size=xy.shape) * sigma
  bad = rng.rand(n) < outliers
  xy[bad] = rng.uniform([0, 0], [640, 480], size=(int(bad.sum()), 2))
  return xyz, xy, R, t, ~bad


def _rot_err_deg(Ra, Rb):
  c = (np.trace(Ra.T @ Rb) - 1) / 2
  return np.degrees(np.arccos(np.clip(c, -1, 1)))


def test_cv_rng_recurrence():
  ref = epnp_numpy.CvRng()
  assert epnp_ref.rng_sequence(50) == [ref.next() for _ in range(50)]


@pytest.mark.parametrize('n', [3, 12])
def test_jacobi_against_lapack(n):
  rng = np.random.RandomState(n)
  for trial in range(5):
    B = rng.normal(size=(n + 3, n)) * (10.0 ** rng.uniform(-3, 3, size=n))
    A = B.T @ B
    if trial == 4:                          # rank deficient, like the 5-point M^T M
      B[:, -2:] = B[:, :2]
      A = B.T @ B
    w, V = epnp_ref.jacobi(A)
    order = np.argsort(w)
    scale = np.abs(A).max()
    np.testing.assert_allclose(w[order], np.linalg.eigvalsh(A), atol=1e-12 * scale)
    np.testing.assert_allclose(V.T @ V, np.eye(n), atol=1e-13)
    np.testing.assert_allclose(A @ V, V * w, atol=1e-11 * scale)


@pytest.mark.parametrize('n', [5, 6, 8, 40, 1000])
@pytest.mark.parametrize('order', [1, 256])
def test_epnp_recovers_a_noise_free_pose(n, order):
  for seed in range(4):
    xyz, xy, R, t, _ = scene(seed + 10 * n, n)
    P = epnp_ref.epnp(xyz, xy, K, order=order)
    assert P is not None
    Y = xyz @ P[:, :3].T + P[:, 3]
    uv = np.stack([K[0, 0] * Y[:, 0] / Y[:, 2] + K[0, 2], K[1, 1] * Y[:, 1] / Y[:, 2] + K[1, 2]], 1)
    assert np.abs(uv - xy).max() < (1e-4 if n >= 6 else 1e-2)
    if n >= 6:
      assert _rot_err_deg(P[:, :3], R) < 1e-4
      assert np.linalg.norm(P[:, 3] - t) < 1e-2
    np.testing.assert_allclose(P[:, :3] @ P[:, :3].T, np.eye(3), atol=1e-12)
    assert np.linalg.det(P[:, :3]) > 0


@pytest.mark.parametrize('n', [6, 10, 60, 600])
def test_epnp_agrees_with_the_numpy_statement_under_noise(n):
  for seed in range(6):
    xyz, xy, R, t, _ = scene(100 + seed + n, n, sigma=1.0)
    P = epnp_ref.epnp(xyz, xy, K, order=256)
    Rn, tn = epnp_numpy.epnp(xyz, xy, K)
    assert P is not None and Rn is not None
    assert _rot_err_deg(P[:, :3], Rn) < 1e-5, (n, seed)
    assert np.linalg.norm(P[:, 3] - tn) < 1e-4 * np.linalg.norm(tn)
    if n >= 60:
      assert _rot_err_deg(P[:, :3], R) < 1.0


@pytest.mark.parametrize('n,outliers', [(200, 0.3), (800, 0.5), (50, 0.0), (3000, 0.6)])
def test_ransac_loop_against_the_numpy_statement(n, outliers):
  """Sampling (cv::RNG, distinct indices), the float32 inlier rule and the shrinking
  iteration bound, restated in numpy around the ORACLE's EPnP for the minimal sets: the
  best set, its inlier count, the bound and the mask must match exactly."""
  xyz, xy, R, t, good = scene(7 + n, n, sigma=1.0, outliers=outliers)
  ok, P, mask, info = epnp_ref.solvePnPRansac(xyz, xy, K, 400, 4.0, 0.99)
  assert ok

  def solver(x3, x2):
    Q = epnp_ref.epnp(x3, x2, K, order=1)
    return (None, None) if Q is None else (Q[:, :3], Q[:, 3])
  best_it, best, niters, evaluated, m = epnp_numpy.ransac_trace(xyz, xy, K, solver)
  assert list(info) == [best_it, best, niters, evaluated]
  assert (mask.astype(bool) == m).all()
  assert _rot_err_deg(P[:, :3], R) < 1.0 and np.linalg.norm(P[:, 3] - t) < 0.02 * t[2]
  assert (mask.astype(bool) & good).sum() >= 0.9 * good.sum()
  if outliers == 0.0:
    assert info[2] <= 2 and info[3] <= 2        # w = 1: log(0.01) / log(1 - 1) -> one set


def test_ransac_final_pose_is_epnp_of_the_inliers():
  xyz, xy, R, t, good = scene(3, 500, sigma=1.0, outliers=0.4)
  ok, P, mask, info = epnp_ref.solvePnPRansac(xyz, xy, K)
  m = mask.astype(bool)
  x32, y32 = xyz.astype(np.float32).astype(np.float64), xy.astype(np.float32).astype(np.float64)
  us = epnp_numpy.solver_image_points(xy.astype(np.float32)[m], K)   # what the solver sees
  Q = epnp_ref.epnp(x32[m], us, K, order=256)
  np.testing.assert_array_equal(P, Q)
  Rn, tn = epnp_numpy.epnp(x32[m], us, K)
  assert _rot_err_deg(P[:, :3], Rn) < 1e-5


def test_ransac_degenerate_inputs():
  xyz, xy, R, t, _ = scene(5, 4)
  ok, P, mask, info = epnp_ref.solvePnPRansac(xyz, xy, K)
  assert not ok and P is None and mask.sum() == 0
  xyz, xy, R, t, _ = scene(6, 5)
  ok, P, mask, info = epnp_ref.solvePnPRansac(xyz, xy, K)
  assert ok and mask.sum() == 5
  rng = np.random.RandomState(0)                  # pure clutter: no set reaches 5 inliers
  ok, P, mask, info = epnp_ref.solvePnPRansac(rng.uniform(-50, 50, (300, 3)),
                                              rng.uniform(0, 480, (300, 2)), K)
  assert info[3] == 400
  assert ok == (info[0] >= 0)


def test_fifth_power_is_rounded_once_and_solver_points_make_the_float32_round_trip():
  """Round 4, the two gaps to OpenCV's published arithmetic that CAN be closed: (i)
  RANSACUpdateNumIters calls std::pow(1 - ep, 5) -- the oracle's w^5 equals the exactly
  computed fifth power rounded once (Fraction arithmetic), where (w*w)*(w*w)*w is an ulp off
  for a good share of the inputs; (ii) solvePnP hands EPnP the image points after
  undistortPoints' float32 output of the NORMALISED coordinate: (float)((u - cx) / fx) mapped
  back with x * fu + uc."""
  import fractions
  rng = np.random.RandomState(0)
  ws = np.concatenate([rng.uniform(0, 1, 4000), np.arange(0, 1001) / 1000.0,
                       1.0 - np.arange(1, 400) / np.arange(401, 800)])
  off = 0
  for w in ws:
    exact = float(fractions.Fraction(float(w)) ** 5)
    assert epnp_ref.pow5(w) == exact, w
    off += ((w * w) * (w * w) * w) != exact
  assert off > 50                                 # the four-rounding product is NOT that
  fu, uc = K[0][0], K[0][2]
  for u in rng.uniform(0, 640, 500).astype(np.float32):
    xn = np.float32((np.float64(u) - uc) * (1.0 / fu))
    assert epnp_ref.us_of(float(u), uc, fu) == float(np.float64(xn) * fu + uc)
    assert abs(epnp_ref.us_of(float(u), uc, fu) - float(u)) < 1e-4   # a few 1e-5 px at most
