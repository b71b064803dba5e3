"""ORACLE (test infrastructure, not product): ctypes binding of oracle/pnp_ref.c
(built by `make -C oracle` / __graft_entry__.build()). Mirrors the call contract of
pyprogressivex.find6DPoses (scripts/infer.py:470-488). PARITY UNPINNED -- see the
header of pnp_ref.c."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libpnp_ref.so')


class PnpRefParams(ctypes.Structure):
  _fields_ = [
      ('threshold', ctypes.c_double),
      ('neighborhood_ball_radius', ctypes.c_double),
      ('spatial_coherence_weight', ctypes.c_double),
      ('scaling_from_millimeters', ctypes.c_double),
      ('max_tanimoto_similarity', ctypes.c_double),
      ('conf', ctypes.c_double),
      ('proposal_engine_conf', ctypes.c_double),
      ('min_coverage', ctypes.c_double),
      ('min_triangle_area', ctypes.c_double),
      ('max_iters', ctypes.c_int32),
      ('min_point_number', ctypes.c_int32),
      ('max_model_number', ctypes.c_int32),
      ('max_model_number_for_optimization', ctypes.c_int32),
      ('use_prosac', ctypes.c_int32),
      ('lo_iters', ctypes.c_int32),
      ('gc_sweeps', ctypes.c_int32),
      ('pearl_iters', ctypes.c_int32),
  ]


_lib = None


def build():
  subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
  global _lib
  if _lib is None:
    src = os.path.join(_HERE, 'pnp_ref.c')
    if not os.path.exists(_LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(_LIB_PATH):
      build()
    _lib = ctypes.CDLL(_LIB_PATH)
    _lib.pnp_ref_find6d_poses.restype = ctypes.c_int
    _lib.pnp_ref_p3p.restype = ctypes.c_int
    _lib.pnp_ref_cubic_root.restype = ctypes.c_double
    _lib.pnp_ref_cubic_root.argtypes = [ctypes.c_double] * 3
  return _lib


def default_params(**kw):
  p = PnpRefParams()
  lib().pnp_ref_params_default(ctypes.byref(p))
  for k, v in kw.items():
    setattr(p, k, v)
  return p


def _ptr(a, t):
  return a.ctypes.data_as(ctypes.POINTER(t))


def find6DPoses(x1y1, x2y2z2, K, params=None, seed=0, max_k=8):
  """Returns (poses [3k,4] or None, labels i32[n], scores f64[k])."""
  xy = np.ascontiguousarray(x1y1, np.float64)
  xyz = np.ascontiguousarray(x2y2z2, np.float64)
  Kd = np.ascontiguousarray(K, np.float64).reshape(9)
  n = xy.shape[0]
  p = params or default_params()
  poses = np.zeros((max_k, 12), np.float64)
  labels = np.full(max(n, 1), -1, np.int32)
  scores = np.zeros(max_k, np.float64)
  k = lib().pnp_ref_find6d_poses(
      _ptr(xy, ctypes.c_double), _ptr(xyz, ctypes.c_double), ctypes.c_int64(n),
      _ptr(Kd, ctypes.c_double), ctypes.byref(p), ctypes.c_uint64(seed),
      _ptr(poses, ctypes.c_double), _ptr(labels, ctypes.c_int32),
      _ptr(scores, ctypes.c_double), ctypes.c_int32(max_k))
  if k <= 0:
    return None, labels[:n], scores[:0]
  out = np.zeros((3 * k, 4))
  for i in range(k):
    out[3 * i:3 * i + 3, :3] = poses[i, :9].reshape(3, 3)
    out[3 * i:3 * i + 3, 3] = poses[i, 9:]
  return out, labels[:n], scores[:k]


def p3p(bearings, points):
  f = np.ascontiguousarray(bearings, np.float64).reshape(9)
  X = np.ascontiguousarray(points, np.float64).reshape(9)
  out = np.zeros(48)
  n = lib().pnp_ref_p3p(_ptr(f, ctypes.c_double), _ptr(X, ctypes.c_double),
                        _ptr(out, ctypes.c_double))
  return [(out[12 * i:12 * i + 9].reshape(3, 3), out[12 * i + 9:12 * i + 12])
          for i in range(n)]


def cubic_root(b, c, d):
  return lib().pnp_ref_cubic_root(b, c, d)


def gc_label(pose34, x1y1, x2y2z2, K, params=None):
  """Spatial-coherence labelling (pnp_ref.c gc_label) of all correspondences under the
  pose [R|t] (3x4): uint8[n] of 0 / 1."""
  xy = np.ascontiguousarray(x1y1, np.float64)
  xyz = np.ascontiguousarray(x2y2z2, np.float64)
  Kd = np.ascontiguousarray(K, np.float64).reshape(9)
  P = np.asarray(pose34, np.float64)
  pose = np.ascontiguousarray(np.concatenate([P[:, :3].reshape(9), P[:, 3]]))
  p = params or default_params()
  n = xy.shape[0]
  out = np.zeros(max(n, 1), np.uint8)
  lib().pnp_ref_gc_label(_ptr(pose, ctypes.c_double), _ptr(Kd, ctypes.c_double),
                         _ptr(xy, ctypes.c_double), _ptr(xyz, ctypes.c_double),
                         ctypes.c_int64(n), ctypes.byref(p), _ptr(out, ctypes.c_uint8))
  return out[:n]
