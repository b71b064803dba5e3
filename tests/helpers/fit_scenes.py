"""Synthetic correspondence sets shaped like what EPOS hands to the fitting stage
(scripts/infer.py:412-488): pixels on the stride-4 head grid in raster order, several
3D candidates per pixel (many-to-many: the true surface point, its symmetric
counterpart, outliers), known poses. Pure numpy; shared by the CPU and GPU tests."""
import numpy as np

K_YCBV = np.array([[1066.8, 0, 313.0], [0, 1067.5, 241.3], [0, 0, 1]])


def rand_rot(rng):
  q = rng.standard_normal(4)
  q /= np.linalg.norm(q)
  w, x, y, z = q
  return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                   [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                   [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def render_ellipsoid(R, t, K, axes, stride=4, h=480, w=640):
  """Pixels (centres of the stride grid, misc.py:26) whose ray hits the ellipsoid
  (X/a)^2 + (Y/b)^2 + (Z/c)^2 = 1 posed by [R|t]; returns (xy [m,2], X_obj [m,3])."""
  ys, xs = np.meshgrid(np.arange(h // stride), np.arange(w // stride), indexing='ij')
  px = np.stack([(xs.ravel() + 0.5) * stride, (ys.ravel() + 0.5) * stride], 1)
  rays = np.linalg.solve(K, np.concatenate([px, np.ones((len(px), 1))], 1).T).T
  d = rays @ R                       # R^T ray, rows
  o = -(R.T @ t)
  inv = 1.0 / np.asarray(axes, np.float64)
  dd, oo = d * inv, o * inv
  a = (dd * dd).sum(1); b = 2 * (dd * oo).sum(1); c = (oo * oo).sum() - 1.0
  disc = b * b - 4 * a * c
  hit = disc > 0
  s = (-b[hit] - np.sqrt(disc[hit])) / (2 * a[hit])
  ok = s > 0
  X = o + d[hit][ok] * s[ok, None]
  return px[hit][ok], X


def dense_scene(rng, instances, axes=(45.0, 30.0, 60.0), sigma3d=1.0, sym=0.5,
                outlier=0.25, K=K_YCBV):
  """instances: list of (R, t). Per visible pixel: the true object point (+ noise), with
  probability `sym` also its counterpart under the 180 deg symmetry about Z (a second,
  equally valid pose explains those), with probability `outlier` a random point of the
  bounding box. Raster order (y, then x, then candidate), as corresp.py:52,67."""
  rows = []
  S = np.diag([-1.0, -1.0, 1.0])
  for inst, (R, t) in enumerate(instances):
    px, X = render_ellipsoid(R, t, K, axes)
    for i in range(len(px)):
      rows.append((px[i, 1], px[i, 0], 0, px[i], X[i] + rng.standard_normal(3) * sigma3d, inst))
      if rng.uniform() < sym:
        rows.append((px[i, 1], px[i, 0], 1, px[i], S @ X[i] + rng.standard_normal(3) * sigma3d, inst))
      if rng.uniform() < outlier:
        rows.append((px[i, 1], px[i, 0], 2, px[i], rng.uniform(-1, 1, 3) * axes, -1))
  rows.sort(key=lambda r: (r[0], r[1], r[2]))
  xy = np.array([r[3] for r in rows], np.float64)
  xyz = np.array([r[4] for r in rows], np.float64)
  src = np.array([r[5] for r in rows], np.int32)
  kind = np.array([r[2] for r in rows], np.int32)
  return xy, xyz, src, kind


def pose_err_sym(R, t, Rg, tg):
  """Rotation error [deg] up to the 180 deg symmetry about the object's Z axis,
  translation error."""
  S = np.diag([-1.0, -1.0, 1.0])
  best = 180.0
  for Q in (np.eye(3), S):
    c = (np.trace(R @ (Rg @ Q).T) - 1) / 2
    best = min(best, float(np.degrees(np.arccos(np.clip(c, -1, 1)))))
  return best, float(np.linalg.norm(np.asarray(t).ravel() - np.asarray(tg).ravel()))


def reproj_residuals(R, t, K, xy, xyz):
  Y = xyz @ np.asarray(R).T + np.asarray(t).ravel()
  p = Y @ np.asarray(K).T
  return p[:, :2] / p[:, 2:] - xy, Y[:, 2]


def reproj_cost_gradient(R, t, K, xy, xyz):
  """Gradient (6-vector: rotation about the camera axes, translation) of
  0.5 * sum ||pi(K (R X + t)) - x||^2, by central differences in fp64 -- an
  implementation that shares nothing with the solver under test."""
  def cost(w, dt):
    th = np.linalg.norm(w)
    if th > 0:
      k = w / th
      Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
      dR = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    else:
      dR = np.eye(3)
    r, _ = reproj_residuals(dR @ R, dR @ np.asarray(t).ravel() + dt, K, xy, xyz)
    return 0.5 * float((r * r).sum())
  g = np.zeros(6)
  for i in range(6):
    e = np.zeros(6)
    h = 1e-6 if i < 3 else 1e-4
    e[i] = h
    g[i] = (cost(e[:3], e[3:]) - cost(-e[:3], -e[3:])) / (2 * h)
  return g


def newton_step_to_stationary_point(R, t, K, xy, xyz):
  """(step [6], predicted cost decrease, cost): the Newton step from the pose to the
  stationary point of 0.5 * sum ||residual||^2 over the given points, with gradient AND
  Hessian from central differences (nothing shared with the solver under test)."""
  R = np.asarray(R, np.float64)
  t = np.asarray(t, np.float64).ravel()

  def moved(x):
    w, dt = x[:3], x[3:]
    th = np.linalg.norm(w)
    if th > 0:
      k = w / th
      Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
      dR = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    else:
      dR = np.eye(3)
    return dR @ R, dR @ t + dt
  hs = np.array([1e-5, 1e-5, 1e-5, 1e-3, 1e-3, 1e-3])
  g = reproj_cost_gradient(R, t, K, xy, xyz)
  H = np.zeros((6, 6))
  for i in range(6):
    e = np.zeros(6)
    e[i] = hs[i]
    Rp, tp = moved(e)
    Rm, tm = moved(-e)
    H[:, i] = (reproj_cost_gradient(Rp, tp, K, xy, xyz) -
               reproj_cost_gradient(Rm, tm, K, xy, xyz)) / (2 * hs[i])
  H = 0.5 * (H + H.T)
  step = -np.linalg.solve(H, g)
  r, _ = reproj_residuals(R, t, K, xy, xyz)
  return step, float(0.5 * g @ np.linalg.solve(H, g)), 0.5 * float((r * r).sum())
