"""CPU restatement of project_to_surface (TEST INFRASTRUCTURE, see oracle/__init__.py):
closest point of a triangle mesh for every query point -- what
``ObjectModelStore.project_pts_to_model`` (datagen.py:128-154) asks of libigl's
``AABB::squared_distance`` and corresp.py:87-88 applies to the predicted 3D points.

PARITY UNPINNED: libigl is an external dependency that is not in the reference tree
and not installable here; its AABB query returns the exact closest point, which this
exhaustive sweep also does (tie between equally close faces -> lowest face index, which
libigl does not specify). Closest point on a triangle: Voronoi-region tests of Ericson,
"Real-Time Collision Detection" 5.1.5, in float64 with a fixed operation order that the
HIP kernel repeats.
"""
import numpy as np


def _dot(a, b):
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def closest_on_triangle(p, a, b, c):
  ab = b - a; ac = c - a; ap = p - a
  d1 = _dot(ab, ap); d2 = _dot(ac, ap)
  if d1 <= 0.0 and d2 <= 0.0:
    return a.copy()
  bp = p - b
  d3 = _dot(ab, bp); d4 = _dot(ac, bp)
  if d3 >= 0.0 and d4 <= d3:
    return b.copy()
  vc = d1 * d4 - d3 * d2
  if vc <= 0.0 and d1 >= 0.0 and d3 <= 0.0:
    v = d1 / (d1 - d3)
    return a + v * ab
  cp = p - c
  d5 = _dot(ab, cp); d6 = _dot(ac, cp)
  if d6 >= 0.0 and d5 <= d6:
    return c.copy()
  vb = d5 * d2 - d1 * d6
  if vb <= 0.0 and d2 >= 0.0 and d6 <= 0.0:
    w = d2 / (d2 - d6)
    return a + w * ac
  va = d3 * d6 - d5 * d4
  if va <= 0.0 and (d4 - d3) >= 0.0 and (d5 - d6) >= 0.0:
    w = (d4 - d3) / ((d4 - d3) + (d5 - d6))
    return b + w * (c - b)
  denom = 1.0 / (va + vb + vc)
  v = vb * denom; w = vc * denom
  return a + ab * v + ac * w


def project_pts_to_model(pts, verts, faces):
  """Returns (closest points f64[N,3], face index int32[N])."""
  pts = np.asarray(pts, np.float64); verts = np.asarray(verts, np.float64)
  faces = np.asarray(faces, np.int64)
  out = np.zeros_like(pts); idx = np.zeros(len(pts), np.int32)
  for i, p in enumerate(pts):
    best, bq, bf = np.inf, None, -1
    for f, (ia, ib, ic) in enumerate(faces):
      q = closest_on_triangle(p, verts[ia], verts[ib], verts[ic])
      d = p - q
      d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
      if d2 < best:
        best, bq, bf = d2, q, f
    out[i] = bq; idx[i] = bf
  return out, idx
