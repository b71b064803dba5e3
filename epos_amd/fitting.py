"""Pose fitting on MI355X -- drop-in for the ``pyprogressivex`` module the
reference imports (scripts/infer.py:470-488; un-vendored danini/progressive-x).

``find6DPoses`` keeps the reference call's keyword names, input layout (C-contiguous
float64 ``x1y1 [n,2]`` / ``x2y2z2 [n,3]``, ``K [3,3]``) and return convention:
``(poses [3k,4] or None, labels, scores)``. The work runs in the HIP kernels of
csrc/pnp_ransac.hip through the C ABI ``epos_find6d_poses``; there is no CPU path.

``solvePnPRansac`` / ``Rodrigues`` stand in for the two cv2 calls of the reference's
alternative fitting method (``--fitting_method=opencv_ransac``, infer.py:505-528).
"""
import ctypes

import numpy as np

from epos_amd import _lib


def fit_params(threshold=4.0, neighborhood_ball_radius=20.0,
               spatial_coherence_weight=0.1, scaling_from_millimeters=0.1,
               max_tanimoto_similarity=0.9, max_iters=400, conf=0.5,
               proposal_engine_conf=1.0, min_coverage=0.5, min_triangle_area=0.0,
               min_point_number=6, max_model_number=1,
               max_model_number_for_optimization=5, use_prosac=False,
               lo_iters=8, gc_sweeps=None, pearl_iters=None):
  """EposFitParams with the reference call's keyword names (infer.py:470-488) plus
  the build's own knobs: lo_iters, gc_sweeps (relabelling sweeps of the
  spatial-coherence step; 0 = off), pearl_iters (joint refinement; 0 = off)."""
  p = _lib.FitParams()
  _lib.load().epos_fit_params_default(ctypes.byref(p))
  p.threshold = threshold
  p.neighborhood_ball_radius = neighborhood_ball_radius
  p.spatial_coherence_weight = spatial_coherence_weight
  p.scaling_from_millimeters = scaling_from_millimeters
  p.max_tanimoto_similarity = max_tanimoto_similarity
  p.max_iters = int(max_iters)
  p.conf = conf
  p.proposal_engine_conf = proposal_engine_conf
  p.min_coverage = min_coverage
  p.min_triangle_area = min_triangle_area
  p.min_point_number = int(min_point_number)
  p.max_model_number = int(max_model_number)
  p.max_model_number_for_optimization = int(max_model_number_for_optimization)
  p.use_prosac = int(bool(use_prosac))
  p.lo_iters = int(lo_iters)
  if gc_sweeps is not None:
    p.gc_sweeps = int(gc_sweeps)
  if pearl_iters is not None:
    p.pearl_iters = int(pearl_iters)
  return p


def find6DPoses(x1y1, x2y2z2, K, threshold=4.0, neighborhood_ball_radius=20.0,
                spatial_coherence_weight=0.1, scaling_from_millimeters=0.1,
                max_tanimoto_similarity=0.9, max_iters=400, conf=0.5,
                proposal_engine_conf=1.0, min_coverage=0.5,
                min_triangle_area=0.0, min_point_number=6, max_model_number=1,
                max_model_number_for_optimization=5, use_prosac=False, log=False,
                seed=0, max_poses=16, gc_sweeps=None, pearl_iters=None):
  """Same contract as pyprogressivex.find6DPoses (infer.py:470-488). Extra
  keyword ``seed`` selects the (counter-based) random stream; ``max_poses`` caps
  the number of instances when max_model_number == -1."""
  del log
  xy = np.ascontiguousarray(x1y1, np.float64)
  xyz = np.ascontiguousarray(x2y2z2, np.float64)
  Kd = np.ascontiguousarray(K, np.float64).reshape(9)
  if xy.ndim != 2 or xy.shape[1] != 2 or xyz.ndim != 2 or xyz.shape[1] != 3 \
        or xy.shape[0] != xyz.shape[0]:
    raise ValueError('x1y1 must be [n,2] and x2y2z2 [n,3]')
  n = xy.shape[0]
  p = fit_params(threshold, neighborhood_ball_radius, spatial_coherence_weight,
                 scaling_from_millimeters, max_tanimoto_similarity, max_iters,
                 conf, proposal_engine_conf, min_coverage, min_triangle_area,
                 min_point_number, max_model_number,
                 max_model_number_for_optimization, use_prosac, gc_sweeps=gc_sweeps,
                 pearl_iters=pearl_iters)
  max_k = max_poses if max_model_number < 0 else max(1, min(max_model_number,
                                                            max_poses))
  poses = np.zeros((max_k, 12), np.float64)
  labels = np.full(max(n, 1), -1, np.int32)
  scores = np.zeros(max_k, np.float64)
  vp = ctypes.c_void_p
  k = _lib.check(_lib.load().epos_find6d_poses(
      xy.ctypes.data_as(vp), xyz.ctypes.data_as(vp), n, Kd.ctypes.data_as(vp),
      ctypes.byref(p), seed, poses.ctypes.data_as(vp), labels.ctypes.data_as(vp),
      scores.ctypes.data_as(vp), max_k), 'epos_find6d_poses')
  if k == 0:
    return None, labels[:n], scores[:0]
  out = np.zeros((3 * k, 4), np.float64)
  for i in range(k):
    out[3 * i:3 * i + 3, :3] = poses[i, :9].reshape(3, 3)
    out[3 * i:3 * i + 3, 3] = poses[i, 9:]
  return out, labels[:n], scores[:k]


SOLVEPNP_EPNP = 1          # cv2.SOLVEPNP_EPNP


def Rodrigues(src):
  """cv2.Rodrigues for the two shapes the reference uses (infer.py:526): a rotation
  vector [3] / [3,1] -> R [3,3], or R [3,3] -> the rotation vector [3,1]. Host numpy."""
  a = np.asarray(src, np.float64)
  if a.size == 3:
    r = a.reshape(3)
    th = float(np.linalg.norm(r))
    if th < 1e-300:
      return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) * np.cos(th) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx
  if a.shape != (3, 3):
    raise ValueError('Rodrigues: expected 3 or 3x3 values')
  w = np.array([a[2, 1] - a[1, 2], a[0, 2] - a[2, 0], a[1, 0] - a[0, 1]])
  s, c = 0.5 * np.linalg.norm(w), 0.5 * (np.trace(a) - 1.0)
  th = np.arctan2(s, c)
  if s > 1e-8:
    return (w / (2.0 * s) * th).reshape(3, 1)
  if c > 0:                                            # theta ~ 0
    return (0.5 * w).reshape(3, 1)
  B = 0.5 * (a + np.eye(3))                            # theta ~ pi: R = 2 k k^T - I
  k = np.sqrt(np.maximum(np.diag(B), 0.0))
  i = int(np.argmax(k))
  k = np.where(B[i] < 0, -k, k)
  k[i] = abs(k[i])
  return (k / np.linalg.norm(k) * th).reshape(3, 1)


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs=None,
                   iterationsCount=100, reprojectionError=8.0, confidence=0.99,
                   flags=SOLVEPNP_EPNP, return_pose=False, return_info=False):
  """cv2.solvePnPRansac with flags=cv2.SOLVEPNP_EPNP as the reference calls it
  (scripts/infer.py:510-518), on the GPU through ``epos_solve_pnp_ransac``
  (csrc/epnp_ransac.hip; there is no CPU path). Same keyword names and defaults, same
  result tuple ``(success, rvec [3,1], tvec [3,1], inliers int32[m,1] or None)``.
  ``return_pose=True`` appends the [3,4] pose (R | t) the kernels computed, so that a
  caller can skip the rvec round trip; ``return_info=True`` appends int32[4] = (index of
  the winning minimal set, its inlier count, the final iteration cap, sets evaluated)."""
  if distCoeffs is not None and np.any(np.asarray(distCoeffs) != 0):
    raise NotImplementedError('solvePnPRansac: distortion coefficients are not supported '
                              '(the reference passes None, infer.py:514)')
  if flags != SOLVEPNP_EPNP:
    raise NotImplementedError('solvePnPRansac: only flags=SOLVEPNP_EPNP (infer.py:518)')
  xyz = np.ascontiguousarray(np.asarray(objectPoints, np.float64).reshape(-1, 3))
  xy = np.ascontiguousarray(np.asarray(imagePoints, np.float64).reshape(-1, 2))
  if xy.shape[0] != xyz.shape[0]:
    raise ValueError('objectPoints and imagePoints differ in length')
  Kd = np.ascontiguousarray(cameraMatrix, np.float64).reshape(9)
  n = xy.shape[0]
  p = _lib.PnpRansacParams()
  lib = _lib.load()
  lib.epos_pnp_ransac_params_default(ctypes.byref(p))
  p.iterations_count = int(iterationsCount)
  p.reprojection_error = float(reprojectionError)
  p.confidence = float(confidence)
  pose = np.zeros(12, np.float64)
  mask = np.zeros(max(n, 1), np.uint8)
  info = np.zeros(4, np.int32)
  vp = ctypes.c_void_p
  ok = _lib.check(lib.epos_solve_pnp_ransac(
      xy.ctypes.data_as(vp), xyz.ctypes.data_as(vp), n, Kd.ctypes.data_as(vp),
      ctypes.byref(p), pose.ctypes.data_as(vp), mask.ctypes.data_as(vp),
      info.ctypes.data_as(vp)), 'epos_solve_pnp_ransac')
  if ok:
    P = np.concatenate([pose[:9].reshape(3, 3), pose[9:].reshape(3, 1)], axis=1)
    res = (True, Rodrigues(P[:, :3]), P[:, 3:].copy(),
           np.nonzero(mask[:n])[0].astype(np.int32).reshape(-1, 1))
  else:
    P = None
    res = (False, None, None, None)
  if return_pose:
    res += (P,)
  if return_info:
    res += (info,)
  return res
