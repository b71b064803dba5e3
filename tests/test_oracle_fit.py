"""CPU tests of the fitting oracle (oracle/pnp_ref.c): the pieces added in round 2 --
spatial-coherence labelling, confidence-bounded termination, Progressive-X retries --
against INDEPENDENT numpy restatements of the same definitions, plus checks that share
nothing with the solver (P3P roots satisfy the three distance constraints, an accepted
pose is a stationary point of its inliers' reprojection cost). PARITY UNPINNED vs
progressive-x (absent from the tree); these pin the oracle to its own specification."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import fit_scenes as fs   # noqa: E402

K = fs.K_YCBV


def _numpy_gc_label(R, t, xy, xyz, thr=4.0, rad=20.0, lam=0.1, s=0.1, sweeps=2):
  """GC-RANSAC's labelling energy, minimised by synchronous ICM sweeps, written the
  slow obvious way: explicit energies of both labels per point, float arithmetic on
  the 2^-20 quantised residuals (exact integers in fp64)."""
  Q = float(1 << 20)
  r, z = fs.reproj_residuals(R, t, K, xy, xyz)
  e2 = (r * r).sum(1)
  d = np.minimum(e2 / (1.5 * thr) ** 2, 1.0)
  d[z <= 0] = 1.0
  q = np.floor(d * Q)
  lab = ((e2 < thr * thr) & (z > 0)).astype(np.int64)
  p5 = np.concatenate([xy, s * xyz], 1)
  D2 = ((p5[:, None, :] - p5[None, :, :]) ** 2).sum(-1)
  nb = (D2 <= rad * rad) & ~np.eye(len(xy), dtype=bool)
  for _ in range(sweeps):
    new = lab.copy()
    for p in range(len(xy)):
      js = np.nonzero(nb[p])[0]
      dq = (q[p] + q[js]) / (2 * Q)                    # (d_p + d_q) / 2 per neighbour
      within = q[p] < Q
      e_in = (0.0 if within else (1 - lam)) + lam * np.where(lab[js] == 1, 1 - dq, 1.0).sum()
      e_out = ((1 - lam) * (1 - q[p] / Q) if within else 0.0) + \
          lam * np.where(lab[js] == 0, dq, 1.0).sum()
      new[p] = 1 if e_in < e_out - 1e-12 else 0
    lab = new
  return lab.astype(np.uint8), nb


def test_gc_label_matches_an_independent_numpy_restatement():
  from oracle import pnp_ref
  rng = np.random.RandomState(3)
  R = fs.rand_rot(rng)
  t = np.array([30.0, -20.0, 700.0])
  xy, xyz, src, kind = fs.dense_scene(rng, [(R, t)], sigma3d=2.5, sym=0.3, outlier=0.3)
  keep = rng.choice(len(xy), 700, replace=False)
  keep.sort()
  xy, xyz = xy[keep], xyz[keep]
  # a slightly wrong pose: many residuals close to the threshold
  dR = fs.rand_rot(np.random.RandomState(1))
  w = 0.004
  Rp = (np.eye(3) + w * (dR - dR.T)) @ R
  u, _, vt = np.linalg.svd(Rp)
  Rp = u @ vt
  P = np.concatenate([Rp, (t + [0.6, -0.4, 3.0])[:, None]], 1)
  got = pnp_ref.gc_label(P, xy, xyz, K)
  want, nb = _numpy_gc_label(Rp, P[:, 3], xy, xyz)
  assert nb.sum() > 20 * len(xy)                      # a real neighbourhood graph
  assert np.array_equal(got, want)
  for sw in (1, 4):                                   # any number of sweeps
    assert np.array_equal(
        pnp_ref.gc_label(P, xy, xyz, K, pnp_ref.default_params(gc_sweeps=sw)),
        _numpy_gc_label(Rp, P[:, 3], xy, xyz, sweeps=sw)[0])
  r, _ = fs.reproj_residuals(Rp, P[:, 3], K, xy, xyz)
  thresholded = ((r * r).sum(1) < 16.0).astype(np.uint8)
  assert 0 < (got != thresholded).sum() < len(xy) // 2   # coherence changed some labels
  # lambda = 0 or radius = 0: the thresholded labelling
  for kw in ({'spatial_coherence_weight': 0.0}, {'neighborhood_ball_radius': 0.0},
             {'gc_sweeps': 0}):
    assert np.array_equal(pnp_ref.gc_label(P, xy, xyz, K, pnp_ref.default_params(**kw)),
                          thresholded)


def test_p3p_roots_satisfy_the_three_distance_constraints():
  from oracle import pnp_ref
  rng = np.random.RandomState(0)
  checked = 0
  for _ in range(200):
    X = rng.uniform(-80, 80, (3, 3))
    R = fs.rand_rot(rng)
    t = np.array([rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(400, 1500)])
    Y = X @ R.T + t
    f = Y / np.linalg.norm(Y, axis=1, keepdims=True)
    sols = pnp_ref.p3p(f, X)
    assert len(sols) >= 1
    found = False
    for Rs, ts in sols:
      Ys = X @ Rs.T + ts                              # camera-frame points of the root
      lam = np.linalg.norm(Ys, axis=1)
      # on the three viewing rays ...
      np.testing.assert_allclose(Ys / lam[:, None], f, atol=1e-9)
      # ... at mutual distances equal to the object-frame ones
      for a, b in ((0, 1), (0, 2), (1, 2)):
        assert abs(np.linalg.norm(Ys[a] - Ys[b]) - np.linalg.norm(X[a] - X[b])) < \
            1e-9 * np.linalg.norm(X[a] - X[b]) + 1e-9
      assert abs(np.linalg.det(Rs) - 1) < 1e-6          # a rotation (conditioning-limited)
      found |= np.allclose(Rs, R, atol=1e-6) and np.allclose(ts, t, atol=1e-4)
      checked += 1
    assert found
  assert checked > 300


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_accepted_pose_is_a_stationary_point_of_its_inliers(seed):
  """The local optimisation ends where the Gauss-Newton step stops improving the MSAC
  score: the pose is then (close to) a minimiser of the plain reprojection cost over
  its inliers. Central-difference gradient in numpy, nothing shared with the C code."""
  from oracle import pnp_ref
  rng = np.random.RandomState(10 + seed)
  R = fs.rand_rot(rng)
  t = np.array([rng.uniform(-80, 80), rng.uniform(-60, 60), rng.uniform(600, 1000)])
  # 0.5 mm of noise on the predicted 3D points = ~1 px after projection (tau_r = 4 px)
  xy, xyz, src, kind = fs.dense_scene(rng, [(R, t)], sigma3d=0.5, sym=0.0, outlier=0.3)
  P, lab, sc = pnp_ref.find6DPoses(xy, xyz, K, seed=seed,
                                   params=pnp_ref.default_params(gc_sweeps=0))
  assert P is not None
  inl = lab == 0
  step, dec, cost = fs.newton_step_to_stationary_point(P[:, :3], P[:, 3], K, xy[inl],
                                                       xyz[inl])
  # The local optimisation stops when a Gauss-Newton step no longer raises the MSAC
  # score, so the pose need not sit exactly on the stationary point of its final inlier
  # set -- but it must be within a negligible step of it: < 1e-3 rad (0.06 deg), < 1 mm,
  # and the cost still to be gained there below 3 % of the cost.
  assert np.linalg.norm(step[:3]) < 1e-3 and np.linalg.norm(step[3:]) < 1.0, step
  assert dec < 0.03 * cost, (dec, cost)
  rot, tr = fs.pose_err_sym(P[:, :3], P[:, 3], R, t)
  assert rot < 1.0 and tr < 0.01 * t[2]


def test_ransac_confidence_bound_and_powi():
  from oracle import pnp_ref
  L = pnp_ref.lib()
  L.pnp_ref_powi.restype = __import__('ctypes').c_double
  L.pnp_ref_powi.argtypes = [__import__('ctypes').c_double, __import__('ctypes').c_int64]
  for b, e in ((0.5, 10), (0.999, 400), (1.0, 7), (0.0, 3), (0.3, 0), (0.9, 1)):
    assert abs(L.pnp_ref_powi(b, e) - b ** e) <= 4e-16 * max(b ** e, 1e-300) * max(e, 1)
  rng = np.random.RandomState(4)
  R = fs.rand_rot(rng)
  t = np.array([10.0, 5.0, 800.0])
  xy, xyz, src, kind = fs.dense_scene(rng, [(R, t)], sym=0.0, outlier=0.2)
  full = pnp_ref.find6DPoses(xy, xyz, K, seed=2)
  # with ~80 % inliers, confidence 0.99 needs (1 - 0.8^3)^it <= 0.01 -> a handful of
  # samples: the result is the best of that prefix, i.e. max_iters = prefix gives it too
  early = pnp_ref.find6DPoses(xy, xyz, K, seed=2, params=pnp_ref.default_params(
      proposal_engine_conf=0.99))
  assert early[0] is not None and full[0] is not None
  hit = None
  for it in range(1, 60):
    cut = pnp_ref.find6DPoses(xy, xyz, K, seed=2, params=pnp_ref.default_params(max_iters=it))
    if cut[0] is not None and np.array_equal(cut[0], early[0]) and np.array_equal(cut[1], early[1]):
      hit = it
      break
  assert hit is not None and hit < 40
  rot, tr = fs.pose_err_sym(early[0][:, :3], early[0][:, 3], R, t)
  assert rot < 1.5


def test_progressive_x_retries_a_failed_proposal_only_in_multi_instance_mode():
  """Two instances, the second one weak: with conf high the failed proposals are
  retried (fresh samples), with conf ~ 0 the search stops at the first failure; a
  single-instance search never retries."""
  from oracle import pnp_ref
  rng = np.random.RandomState(8)
  insts = [(fs.rand_rot(rng), np.array([-120.0, 20.0, 750.0])),
           (fs.rand_rot(rng), np.array([140.0, -30.0, 900.0]))]
  xy, xyz, src, kind = fs.dense_scene(rng, insts, sym=0.0, outlier=0.5)
  counts = {}
  for conf in (1e-9, 0.999999):
    n_found = []
    for seed in range(12):
      P, lab, sc = pnp_ref.find6DPoses(
          xy, xyz, K, seed=seed, max_k=4,
          params=pnp_ref.default_params(max_model_number=4, conf=conf, max_iters=12))
      n_found.append(0 if P is None else P.shape[0] // 3)
    counts[conf] = n_found
  assert sum(counts[0.999999]) > sum(counts[1e-9])      # retries found more instances
  assert all(a >= b for a, b in zip(counts[0.999999], counts[1e-9]))


def _labelling_terms(P, xy, xyz, thr=4.0, rad=20.0, s=0.1):
  Q = 1 << 20
  r, z = fs.reproj_residuals(P[:, :3], P[:, 3], K, xy, xyz)
  e2 = (r * r).sum(1)
  d = np.minimum(e2 / (1.5 * thr) ** 2, 1.0)
  d[z <= 0] = 1.0
  q = np.floor(d * Q).astype(np.int64)
  p5 = np.concatenate([xy, s * xyz], 1)
  D2 = ((p5[:, None, :] - p5[None, :, :]) ** 2).sum(-1)
  nb = (D2 <= rad * rad) & ~np.eye(len(xy), dtype=bool)
  return q, nb, ((e2 < thr * thr) & (z > 0)).astype(np.int64)


def _labelling_energy(lab, q, nb):
  """The labelling energy of pnp_ref.c's gc_label for lambda = 0.1, scaled by 2 Q / lambda
  (all integers): unary 18 Q / 18 (Q - q_p), pairwise q_p + q_q | 2 Q | 2 Q - (q_p + q_q)."""
  Q = 1 << 20
  within = q < Q
  U = np.where(lab == 1, np.where(within, 0, 18 * Q), np.where(within, 18 * (Q - q), 0)).sum()
  i, j = np.nonzero(np.triu(nb, 1))
  sq = q[i] + q[j]
  V = np.where((lab[i] == 1) & (lab[j] == 1), 2 * Q - sq,
               np.where((lab[i] == 0) & (lab[j] == 0), sq, 2 * Q)).sum()
  return int(U + V)


def _min_cut_labelling(q, nb):
  """The EXACT minimiser of that energy by an s-t minimum cut (scipy's max-flow; the energy
  is submodular: V(0,0) + V(1,1) = 2 Q <= V(0,1) + V(1,0) = 4 Q). Source side = outlier.
  Returns the minimal-inlier-set solution (nodes not reachable from the source in the
  residual graph are inliers)."""
  import scipy.sparse as sp
  from scipy.sparse.csgraph import breadth_first_order, maximum_flow
  Q = 1 << 20
  n = len(q)
  within = q < Q
  th1 = np.where(within, 0, 18 * Q).astype(np.int64)          # cost of the inlier label
  th0 = np.where(within, 18 * (Q - q), 0).astype(np.int64)    # cost of the outlier label
  i, j = np.nonzero(np.triu(nb, 1))
  sq = q[i] + q[j]
  # theta(x_i, x_j) = A + (C - A) x_i + (D - C) x_j + (B + C - A - D) [x_i = 0, x_j = 1]
  A, B, C, D = sq, 2 * Q, 2 * Q, 2 * Q - sq
  np.add.at(th1, i, C - A)
  np.add.at(th1, j, D - C)
  diff = th1 - th0
  pos, neg = np.nonzero(diff > 0)[0], np.nonzero(diff < 0)[0]
  src, snk = n, n + 1
  rows = np.concatenate([i, np.full(len(pos), src), neg])
  cols = np.concatenate([j, pos, np.full(len(neg), snk)])
  vals = np.concatenate([np.full(len(i), B + C, np.int64) - A - D, diff[pos], -diff[neg]])
  G = sp.csr_matrix((vals.astype(np.int32), (rows, cols)), shape=(n + 2, n + 2))
  flow = maximum_flow(G, src, snk).flow
  resid = (G - flow).maximum(0) + flow.T.maximum(0)
  resid.eliminate_zeros()
  reach = np.zeros(n + 2, bool)
  reach[breadth_first_order(resid, src, directed=True, return_predecessors=False)] = True
  return (~reach[:n]).astype(np.int64)


def _near_pose_scene(seed, n_keep=500):
  rng = np.random.RandomState(seed)
  R = fs.rand_rot(rng)
  t = np.array([30.0, -20.0, 700.0])
  xy, xyz, _, _ = fs.dense_scene(rng, [(R, t)], sigma3d=2.5, sym=0.3, outlier=0.3)
  keep = np.sort(rng.choice(len(xy), n_keep, replace=False))
  dR = fs.rand_rot(np.random.RandomState(seed + 1))
  u, _, vt = np.linalg.svd((np.eye(3) + 0.004 * (dR - dR.T)) @ R)
  return np.concatenate([u @ vt, (t + [0.6, -0.4, 3.0])[:, None]], 1), xy[keep], xyz[keep]


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_gc_label_against_the_exact_minimum_cut(seed):
  """How good is "synchronous sweeps instead of an exact s-t min-cut"? Checked against
  the exact minimiser of the same energy (max-flow):
  * on SPARSE neighbourhood graphs (a few neighbours per point) the default two sweeps ARE
    the minimum cut -- every label equal (one sweep: within 1 % of the minimum energy);
  * on the DENSE graphs EPOS's many-to-many candidates give at tau_d = 20 (dozens of
    neighbours) the exact minimiser is (nearly) degenerate -- it labels almost every point an
    inlier, because lambda x degree outweighs the unary term -- and the sweeps stay between
    the thresholded labelling and it in energy. There the number of sweeps is a parameter
    of the method, not a convergence knob."""
  from oracle import pnp_ref
  P, xy, xyz = _near_pose_scene(seed)
  for rad in (5.0, 8.0):                                # sparse: exact
    q, nb, thr = _labelling_terms(P, xy, xyz, rad=rad)
    assert 0.5 < nb.sum() / len(q) < 12
    exact = _min_cut_labelling(q, nb)
    for sweeps in (1, 2, 8):
      got = pnp_ref.gc_label(P, xy, xyz, K, pnp_ref.default_params(
          gc_sweeps=sweeps, neighborhood_ball_radius=rad)).astype(np.int64)
      e_got, e_min = _labelling_energy(got, q, nb), _labelling_energy(exact, q, nb)
      if sweeps == 1:                                    # one sweep: within 1 % and 2 labels
        assert e_min <= e_got <= 1.01 * e_min and (got != exact).sum() <= 2, (rad, sweeps)
      else:                                              # the default two sweeps: THE minimum
        assert e_got == e_min and np.array_equal(got, exact), (rad, sweeps)
    assert _labelling_energy(thr, q, nb) > _labelling_energy(exact, q, nb)
  q, nb, thr = _labelling_terms(P, xy, xyz, rad=20.0)  # dense: bounded by the two
  assert nb.sum() / len(q) > 25
  exact = _min_cut_labelling(q, nb)
  e_exact = _labelling_energy(exact, q, nb)
  assert exact.sum() >= 0.9 * len(q) and thr.sum() < 0.5 * len(q)
  for sweeps in (1, 2, 4):
    got = pnp_ref.gc_label(P, xy, xyz, K, pnp_ref.default_params(gc_sweeps=sweeps)).astype(np.int64)
    assert e_exact <= _labelling_energy(got, q, nb)


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_exact_mode_of_the_c_oracle_is_the_minimum_cut(seed):
  """gc_sweeps < 0 selects pnp_ref.c's own exact mode (gc_label_exact: Dinic max-flow on the
  integer energy). Pinned here against scipy's max-flow on graphs from ~1 to ~40 neighbours
  per point (same labels: both return the minimum cut with the smallest source side), and
  used to state where the product's two sweeps ARE the exact answer: every graph of at
  most 5 neighbours per point on average."""
  from oracle import pnp_ref
  P, xy, xyz = _near_pose_scene(seed)
  for rad in (5.0, 8.0, 12.0, 20.0):
    q, nb, _ = _labelling_terms(P, xy, xyz, rad=rad)
    exact = pnp_ref.gc_label(P, xy, xyz, K, pnp_ref.default_params(
        gc_sweeps=-1, neighborhood_ball_radius=rad)).astype(np.int64)
    assert np.array_equal(exact, _min_cut_labelling(q, nb)), rad
    sweeps = pnp_ref.gc_label(P, xy, xyz, K, pnp_ref.default_params(
        gc_sweeps=2, neighborhood_ball_radius=rad)).astype(np.int64)
    if nb.sum() / len(q) <= 5.0:
      assert np.array_equal(sweeps, exact), rad
    else:
      assert _labelling_energy(exact, q, nb) <= _labelling_energy(sweeps, q, nb)


def test_exact_mode_handles_degenerate_graphs():
  """No neighbours at all (radius below the pixel pitch): the cut is the unary decision
  q < Q; one point; every point behind the camera."""
  from oracle import pnp_ref
  P, xy, xyz = _near_pose_scene(0, n_keep=60)
  prm = pnp_ref.default_params(gc_sweeps=-1, neighborhood_ball_radius=0.5)
  q, nb, _ = _labelling_terms(P, xy, xyz, rad=0.5)
  assert nb.sum() == 0
  got = pnp_ref.gc_label(P, xy, xyz, K, prm)
  assert np.array_equal(got.astype(bool), q < (1 << 20))
  assert pnp_ref.gc_label(P, xy[:1], xyz[:1], K, prm).shape == (1,)
  back = P.copy()
  back[:, 3] = [0.0, 0.0, -700.0]
  assert not pnp_ref.gc_label(back, xy, xyz, K, pnp_ref.default_params(gc_sweeps=-1)).any()


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_fit_with_two_sweeps_equals_fit_with_exact_cuts_on_sparse_graphs(seed):
  """The whole single-instance fit (proposal -> local optimisation -> labelled refits) with
  the labelling done by two sweeps and by exact minimum cuts: the same labels and poses
  on sparse neighbourhood graphs."""
  from oracle import pnp_ref
  rng = np.random.RandomState(140 + seed)
  R = fs.rand_rot(rng)
  t = np.array([rng.uniform(-100, 100), rng.uniform(-60, 60), rng.uniform(600, 1000)])
  xy, xyz, _, _ = fs.dense_scene(rng, [(R, t)], sigma3d=2.0, sym=0.3, outlier=0.3)
  for rad in (5.0, 8.0):
    a = pnp_ref.find6DPoses(xy, xyz, K, params=pnp_ref.default_params(
        neighborhood_ball_radius=rad, gc_sweeps=2), seed=seed, max_k=1)
    b = pnp_ref.find6DPoses(xy, xyz, K, params=pnp_ref.default_params(
        neighborhood_ball_radius=rad, gc_sweeps=-1), seed=seed, max_k=1)
    assert a[0] is not None and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_fit_at_the_default_radius_equals_fit_with_exact_cuts(seed):
  """At EPOS's default tau_d = 20 the neighbourhood graphs are dense (dozens of neighbours) and
  two sweeps are NOT the minimum cut of the labelling energy -- but the labelling only ever
  feeds refits that are kept when the MSAC quality grows, and what comes out of the whole
  fit is the same: labels and poses with two sweeps equal labels and poses with exact s-t
  minimum cuts, on single-instance scenes over a range of noise / symmetry / outlier levels
  and on a multi-instance search (40 + 10 scenes at the time of writing; a sample here)."""
  from oracle import pnp_ref
  rng = np.random.RandomState(500 + seed)
  R = fs.rand_rot(rng)
  t = np.array([rng.uniform(-100, 100), rng.uniform(-60, 60), rng.uniform(600, 1000)])
  xy, xyz, _, _ = fs.dense_scene(rng, [(R, t)], sigma3d=rng.uniform(0.3, 2.5),
                                 sym=rng.uniform(0, 1), outlier=rng.uniform(0.1, 0.6))
  a = pnp_ref.find6DPoses(xy, xyz, K, params=pnp_ref.default_params(gc_sweeps=2), seed=seed, max_k=1)
  b = pnp_ref.find6DPoses(xy, xyz, K, params=pnp_ref.default_params(gc_sweeps=-1), seed=seed, max_k=1)
  assert a[0] is not None and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
  if seed < 2:
    rng = np.random.RandomState(900 + seed)
    inst = [(fs.rand_rot(rng), np.array([-38.0, 10.0, 760.0])),
            (fs.rand_rot(rng), np.array([42.0, -5.0, 790.0]))]
    xy, xyz, _, _ = fs.dense_scene(rng, inst, sigma3d=0.7, sym=1.0, outlier=0.2)
    a = pnp_ref.find6DPoses(xy, xyz, K, params=pnp_ref.default_params(
        gc_sweeps=2, max_model_number=3), seed=seed, max_k=3)
    b = pnp_ref.find6DPoses(xy, xyz, K, params=pnp_ref.default_params(
        gc_sweeps=-1, max_model_number=3), seed=seed, max_k=3)
    assert a[0].shape == b[0].shape and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
