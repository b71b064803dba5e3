"""Pose fitting on MI355X -- drop-in for the ``pyprogressivex`` module the
reference imports (scripts/infer.py:470-488; un-vendored danini/progressive-x).

``find6DPoses`` keeps the reference call's keyword names, input layout (C-contiguous
float64 ``x1y1 [n,2]`` / ``x2y2z2 [n,3]``, ``K [3,3]``) and return convention:
``(poses [3k,4] or None, labels, scores)``. The work runs in the HIP kernels of
csrc/pnp_ransac.hip through the C ABI ``epos_find6d_poses``; there is no CPU path.
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
