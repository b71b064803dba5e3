"""Independent numpy statement of EPnP (Lepetit, Moreno-Noguer & Fua, IJCV 2009) and of the
RANSAC loop cv2.solvePnPRansac wraps around it, used to check oracle/epnp_ref.c: LAPACK
factorisations (eigh / lstsq / svd), per-point sums -- none of the oracle's shortcuts."""
import fractions

import numpy as np

PAIRS = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]


def _candidates(L, rho):
  out = []
  x = np.linalg.lstsq(L[:, [0, 1, 3, 6]], rho, rcond=None)[0]
  s = -1.0 if x[0] < 0 else 1.0
  b0 = np.sqrt(s * x[0])
  out.append(np.array([b0, s * x[1] / b0, s * x[2] / b0, s * x[3] / b0]))
  x = np.linalg.lstsq(L[:, :3], rho, rcond=None)[0]
  if x[0] < 0:
    b = [np.sqrt(-x[0]), np.sqrt(-x[2]) if x[2] < 0 else 0.0]
  else:
    b = [np.sqrt(x[0]), np.sqrt(x[2]) if x[2] > 0 else 0.0]
  if x[1] < 0:
    b[0] = -b[0]
  out.append(np.array([b[0], b[1], 0.0, 0.0]))
  x = np.linalg.lstsq(L[:, :5], rho, rcond=None)[0]
  if x[0] < 0:
    b = [np.sqrt(-x[0]), np.sqrt(-x[2]) if x[2] < 0 else 0.0]
  else:
    b = [np.sqrt(x[0]), np.sqrt(x[2]) if x[2] > 0 else 0.0]
  if x[1] < 0:
    b[0] = -b[0]
  out.append(np.array([b[0], b[1], x[3] / b[0], 0.0]))
  return out


def _gauss_newton(L, rho, be):
  be = be.copy()
  for _ in range(5):
    b0, b1, b2, b3 = be
    A = np.stack([2 * L[:, 0] * b0 + L[:, 1] * b1 + L[:, 3] * b2 + L[:, 6] * b3,
                  L[:, 1] * b0 + 2 * L[:, 2] * b1 + L[:, 4] * b2 + L[:, 7] * b3,
                  L[:, 3] * b0 + L[:, 4] * b1 + 2 * L[:, 5] * b2 + L[:, 8] * b3,
                  L[:, 6] * b0 + L[:, 7] * b1 + L[:, 8] * b2 + 2 * L[:, 9] * b3], axis=1)
    prod = np.array([b0 * b0, b0 * b1, b1 * b1, b0 * b2, b1 * b2, b2 * b2, b0 * b3, b1 * b3,
                     b2 * b3, b3 * b3])
    be = be + np.linalg.lstsq(A, rho - L @ prod, rcond=None)[0]
  return be


def epnp(xyz, xy, K):
  """-> (R, t) of the candidate with the smallest mean reprojection error."""
  xyz, xy = np.asarray(xyz, np.float64), np.asarray(xy, np.float64)
  n = len(xyz)
  fu, fv, uc, vc = K[0][0], K[1][1], K[0][2], K[1][2]
  c0 = xyz.mean(0)
  d = xyz - c0
  ev, E = np.linalg.eigh(d.T @ d)
  dirs = []
  for j in np.argsort(-ev, kind='stable'):       # the oracle's convention: by decreasing
    e = E[:, j]                                  # eigenvalue, largest component positive
    e = e if e[np.argmax(np.abs(e))] > 0 else -e
    dirs.append(np.sqrt(max(ev[j], 0) / n) * e)
  cws = np.vstack([c0] + [c0 + v for v in dirs])
  CC = (cws[1:] - cws[0]).T
  a123 = np.linalg.solve(CC, d.T).T
  al = np.hstack([1 - a123.sum(1, keepdims=True), a123])
  M = np.zeros((2 * n, 12))
  for j in range(4):
    M[0::2, 3 * j] = al[:, j] * fu
    M[0::2, 3 * j + 2] = al[:, j] * (uc - xy[:, 0])
    M[1::2, 3 * j + 1] = al[:, j] * fv
    M[1::2, 3 * j + 2] = al[:, j] * (vc - xy[:, 1])
  w, V = np.linalg.eigh(M.T @ M)
  v4 = [V[:, k].reshape(4, 3) for k in range(4)]
  L = np.zeros((6, 10))
  rho = np.zeros(6)
  for r, (a, b) in enumerate(PAIRS):
    dv = [v[a] - v[b] for v in v4]
    L[r] = [dv[0] @ dv[0], 2 * dv[0] @ dv[1], dv[1] @ dv[1], 2 * dv[0] @ dv[2],
            2 * dv[1] @ dv[2], dv[2] @ dv[2], 2 * dv[0] @ dv[3], 2 * dv[1] @ dv[3],
            2 * dv[2] @ dv[3], dv[3] @ dv[3]]
    rho[r] = ((cws[a] - cws[b]) ** 2).sum()
  best = None
  with np.errstate(all='ignore'):
    for be in _candidates(L, rho):
      if not np.isfinite(be).all():
        continue
      be = _gauss_newton(L, rho, be)
      ccs = sum(be[k] * v4[k] for k in range(4))
      pcs = al @ ccs
      if pcs[0, 2] < 0:
        ccs, pcs = -ccs, -pcs
      pc0, pw0 = pcs.mean(0), xyz.mean(0)
      U, _, Vt = np.linalg.svd((pcs - pc0).T @ (xyz - pw0))
      R = U @ Vt
      if np.linalg.det(R) < 0:
        R[2] = -R[2]
      t = pc0 - R @ pw0
      Y = xyz @ R.T + t
      ue, ve = uc + fu * Y[:, 0] / Y[:, 2], vc + fv * Y[:, 1] / Y[:, 2]
      err = np.sqrt((xy[:, 0] - ue) ** 2 + (xy[:, 1] - ve) ** 2).mean()
      if np.isfinite(err) and (best is None or err < best[0]):
        best = (err, R, t)
  return (None, None) if best is None else (best[1], best[2])


class CvRng(object):
  """cv::RNG: multiply-with-carry, `state = (uint32)state * 4164903690 + (state >> 32)`."""

  def __init__(self, state=(1 << 64) - 1):
    self.state = state

  def next(self):
    self.state = ((self.state & 0xffffffff) * 4164903690 + (self.state >> 32)) & ((1 << 64) - 1)
    return self.state & 0xffffffff


def solver_image_points(xy32, K):
  """What the EPnP solver sees of float32 image points: solvePnP's undistortPoints writes the
  NORMALISED coordinate as float32, epnp::init_points maps it back with x * fu + uc (double).
  The inlier test keeps the float32 pixels."""
  xy32 = np.asarray(xy32, np.float32)
  fu, fv, uc, vc = K[0][0], K[1][1], K[0][2], K[1][2]
  xn = ((xy32[:, 0].astype(np.float64) - uc) * (1.0 / fu)).astype(np.float32)
  yn = ((xy32[:, 1].astype(np.float64) - vc) * (1.0 / fv)).astype(np.float32)
  return np.stack([xn.astype(np.float64) * fu + uc, yn.astype(np.float64) * fv + vc], 1)


def inlier_mask_f32(R, t, xyz32, xy32, K, thr):
  """PnPRansacCallback::computeError + findInliers: float32 projection, float32 error."""
  fu, fv, uc, vc = K[0][0], K[1][1], K[0][2], K[1][2]
  Y = xyz32.astype(np.float64) @ np.asarray(R).T + np.asarray(t).reshape(1, 3)
  with np.errstate(all='ignore'):
    iz = np.where(Y[:, 2] != 0, 1.0 / Y[:, 2], 1.0)
    up = ((Y[:, 0] * iz) * fu + uc).astype(np.float32)
    vp = ((Y[:, 1] * iz) * fv + vc).astype(np.float32)
    dx = (xy32[:, 0] - up).astype(np.float32)
    dy = (xy32[:, 1] - vp).astype(np.float32)
    err = (dx.astype(np.float64) ** 2 + dy.astype(np.float64) ** 2).astype(np.float32)
  return err <= np.float32(thr * thr)


def ransac_trace(xyz, xy, K, solver, iters=400, thr=4.0, conf=0.99):
  """The loop of RANSACPointSetRegistrator::run around `solver(xyz5, xy5) -> (R, t) or
  (None, None)`: returns (best_it, best_count, niters, evaluated, mask)."""
  xyz32, xy32 = np.asarray(xyz, np.float32), np.asarray(xy, np.float32)
  n = len(xyz32)
  rng = CvRng()
  niters, best, best_it, it, best_mask = iters, 0, -1, 0, None
  while it < niters:
    s = []
    while len(s) < 5:
      c = rng.next() % n
      if c not in s:
        s.append(c)
    R, t = solver(xyz32[s].astype(np.float64), solver_image_points(xy32[s], K))
    if R is not None:
      m = inlier_mask_f32(R, t, xyz32, xy32, K, thr)
      c = int(m.sum())
      if c > max(best, 4):
        best, best_it, best_mask = c, it, m
        ep = min(max((n - c) / n, 0.0), 1.0)
        num = max(1 - min(max(conf, 0.0), 1.0), np.finfo(np.float64).tiny)
        # std::pow(1 - ep, 5): the exactly computed fifth power, rounded once
        den = 1 - float(fractions.Fraction(1 - ep) ** 5)
        if den < np.finfo(np.float64).tiny:
          niters = 0
        else:
          num, den = np.log(num), np.log(den)
          niters = niters if (den >= 0 or -num >= niters * -den) else int(np.rint(num / den))
    it += 1
  return best_it, best, niters, it, best_mask
