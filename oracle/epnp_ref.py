"""ORACLE (test infrastructure, not product): ctypes binding of oracle/epnp_ref.c, the
restatement of cv2.solvePnPRansac(..., flags=cv2.SOLVEPNP_EPNP) as the reference calls it
(scripts/infer.py:505-528). PARITY UNPINNED -- see the header of epnp_ref.c."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libepnp_ref.so')
_lib = None


def lib():
  global _lib
  if _lib is None:
    src = os.path.join(_HERE, 'epnp_ref.c')
    if not os.path.exists(_LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(_LIB_PATH):
      subprocess.check_call(['make', '-s', '-C', _HERE])
    _lib = ctypes.CDLL(_LIB_PATH)
    _lib.epnp_ref_solve_pnp_ransac.restype = ctypes.c_int
    _lib.epnp_ref_epnp.restype = ctypes.c_int
    _lib.epnp_ref_rng_next.restype = ctypes.c_uint32
    _lib.epnp_ref_pow5.restype = ctypes.c_double
    _lib.epnp_ref_pow5.argtypes = [ctypes.c_double]
    _lib.epnp_ref_us_of.restype = ctypes.c_double
    _lib.epnp_ref_us_of.argtypes = [ctypes.c_double] * 3
  return _lib


def _ptr(a, t):
  return a.ctypes.data_as(ctypes.POINTER(t))


def _pose34(p):
  return np.concatenate([p[:9].reshape(3, 3), p[9:].reshape(3, 1)], axis=1)


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, iterationsCount=400,
                   reprojectionError=4.0, confidence=0.99):
  """-> (success, pose [3,4] or None, inlier mask uint8[n], info int32[4])."""
  xyz = np.ascontiguousarray(objectPoints, np.float64)
  xy = np.ascontiguousarray(imagePoints, np.float64)
  K = np.ascontiguousarray(cameraMatrix, np.float64).reshape(9)
  n = xy.shape[0]
  pose = np.zeros(12)
  mask = np.zeros(max(n, 1), np.uint8)
  info = np.zeros(4, np.int32)
  ok = lib().epnp_ref_solve_pnp_ransac(
      _ptr(xy, ctypes.c_double), _ptr(xyz, ctypes.c_double), ctypes.c_int64(n),
      _ptr(K, ctypes.c_double), ctypes.c_int(iterationsCount),
      ctypes.c_double(reprojectionError), ctypes.c_double(confidence),
      _ptr(pose, ctypes.c_double), _ptr(mask, ctypes.c_uint8), _ptr(info, ctypes.c_int32))
  return bool(ok), (_pose34(pose) if ok else None), mask[:n], info


def epnp(objectPoints, imagePoints, cameraMatrix, order=1):
  """EPnP over all points -> pose [3,4] or None. order: 1 (left to right) or 256."""
  xyz = np.ascontiguousarray(objectPoints, np.float64)
  xy = np.ascontiguousarray(imagePoints, np.float64)
  K = np.ascontiguousarray(cameraMatrix, np.float64).reshape(9)
  pose = np.zeros(12)
  bad = lib().epnp_ref_epnp(_ptr(xy, ctypes.c_double), _ptr(xyz, ctypes.c_double),
                            ctypes.c_int64(xy.shape[0]), _ptr(K, ctypes.c_double),
                            ctypes.c_int(order), _ptr(pose, ctypes.c_double))
  return None if bad else _pose34(pose)


def jacobi(A):
  """(eigenvalues, V) of a symmetric matrix by the oracle's cyclic Jacobi."""
  A = np.array(A, np.float64, order='C')
  n = A.shape[0]
  V = np.zeros((n, n))
  lib().epnp_ref_jacobi(ctypes.c_int(n), _ptr(A, ctypes.c_double), _ptr(V, ctypes.c_double))
  return np.diag(A).copy(), V


def rng_sequence(count, state=0xffffffffffffffff):
  st = ctypes.c_uint64(state)
  return [lib().epnp_ref_rng_next(ctypes.byref(st)) for _ in range(count)]


def pow5(w):
  """w^5 rounded once (the oracle's statement of std::pow(w, 5))."""
  return float(lib().epnp_ref_pow5(float(w)))


def us_of(u_f32, c, f):
  """The image coordinate as the EPnP solver sees it (float32 round trip of the normalised
  coordinate)."""
  return float(lib().epnp_ref_us_of(float(u_f32), float(c), float(f)))
