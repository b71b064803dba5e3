#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Imports ``epos_lib.corresp`` and ``epos_lib.fragment`` from /root/reference with
the import-time-only dependencies (tensorflow, cv2) stubbed,
runs them on seeded synthetic inputs and stores inputs + outputs as ``.npz``.
The fixtures are data (inputs and expected outputs); no reference source travels.
"""
import os
import sys
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = '/root/reference'


def import_reference():
  for name in ['tensorflow', 'tensorflow.python', 'tensorflow.python.ops',
               'tensorflow.python.ops.variables', 'cv2']:
    sys.modules[name] = mock.MagicMock()
  sys.path.insert(0, REFERENCE)
  from epos_lib import corresp, fragment  # pylint: disable=import-error
  return corresp, fragment


def softmax(x, axis):
  x = x - x.max(axis=axis, keepdims=True)
  e = np.exp(x)
  return (e / e.sum(axis=axis, keepdims=True)).astype(np.float32)


def synth_heads(rng, h, w, num_objs, num_frags, sharp):
  obj = softmax(rng.standard_normal((h, w, num_objs + 1)) * sharp, 2)
  frag = softmax(rng.standard_normal((h, w, num_objs, num_frags)) * sharp, 3)
  loc = (rng.standard_normal((h, w, num_objs, num_frags, 3)) * 0.5).astype(
      np.float32)
  return obj, frag, loc


def synth_store(rng, num_objs, num_frags):
  centers = {o: rng.uniform(-80, 80, (num_frags, 3)) for o in
             range(1, num_objs + 1)}
  sizes = {o: rng.uniform(5, 40, num_frags) for o in range(1, num_objs + 1)}
  return centers, sizes


class ModelStore(object):
  def __init__(self, obj_ids, centers, sizes):
    self.dp_model = {'obj_ids': list(obj_ids)}
    self.frag_centers = centers
    self.frag_sizes = sizes


def corresp_case(corresp, name, seed, h, w, num_objs, num_frags, gt_obj_ids,
                 only_annotated, sharp=2.0, tie=False, tau_a=0.1, tau_b=0.5,
                 output_scale=0.25):
  rng = np.random.RandomState(seed)
  obj, frag, loc = synth_heads(rng, h, w, num_objs, num_frags, sharp)
  if tie:
    # Exact tie at conf == max * tau_b for object 1 at a pixel that passes tau_a:
    # a strict '>' must drop the tied fragment (corresp.py:64).
    obj[3, 5, :] = 0.0
    obj[3, 5, 1] = 1.0
    frag[3, 5, 0, :] = 0.0
    frag[3, 5, 0, 7] = 0.5
    frag[3, 5, 0, 9] = 0.25       # == 0.5 * 0.5 exactly
    frag[3, 5, 0, 11] = 0.25 + 2 ** -20
  centers, sizes = synth_store(rng, num_objs, num_frags)
  store = ModelStore(range(1, num_objs + 1), centers, sizes)
  out = corresp.establish_many_to_many(
      obj_confs=obj, frag_confs=frag, frag_coords=loc, gt_obj_ids=gt_obj_ids,
      model_store=store, output_scale=output_scale, min_obj_conf=tau_a,
      min_frag_rel_conf=tau_b, project_to_surface=False,
      only_annotated_objs=only_annotated)
  blob = {
      'obj_confs': obj, 'frag_confs': frag, 'frag_coords': loc,
      'gt_obj_ids': np.asarray(gt_obj_ids, np.int64),
      'only_annotated': np.asarray(only_annotated),
      'output_scale': np.asarray(output_scale, np.float64),
      'min_obj_conf': np.asarray(tau_a, np.float64),
      'min_frag_rel_conf': np.asarray(tau_b, np.float64),
      'frag_centers': np.stack([centers[o] for o in range(1, num_objs + 1)]),
      'frag_sizes': np.stack([sizes[o] for o in range(1, num_objs + 1)]),
      'out_obj_ids': np.asarray(sorted(out.keys()), np.int64),
  }
  for oid, d in out.items():
    for k, v in d.items():
      blob['out_%d_%s' % (oid, k)] = v
  path = os.path.join(HERE, 'corresp_%s.npz' % name)
  np.savez_compressed(path, **blob)
  print(path, {o: len(d['px_id']) for o, d in out.items()})


def fragment_case(fragment, name, seed, num_pts, num_frags):
  rng = np.random.RandomState(seed)
  radii = rng.uniform(30, 80, 3)
  d = rng.standard_normal((num_pts, 3))
  pts = d / np.linalg.norm(d, axis=1, keepdims=True) * radii
  centers, ids = fragment.fragmentation_fps(pts.copy(), num_frags)
  path = os.path.join(HERE, 'fragment_%s.npz' % name)
  np.savez_compressed(path, vertices=pts, num_frags=np.asarray(num_frags),
                      frag_centers=centers, vertex_frag_ids=ids)
  print(path, centers.shape)


def main():
  corresp, fragment = import_reference()
  # name, seed, h, w, O, F, gt ids, only_annotated
  corresp_case(corresp, 'o1_s0', 0, 30, 40, 1, 64, [1], True)
  corresp_case(corresp, 'o3_s1', 1, 20, 24, 3, 64, [1, 3], True)
  corresp_case(corresp, 'o3_all_s2', 2, 20, 24, 3, 64, [2], False)
  corresp_case(corresp, 'o21_s3', 3, 10, 12, 21, 64, [2, 5, 9, 14, 21], True,
               sharp=3.0)
  corresp_case(corresp, 'o3_tie_s4', 4, 12, 16, 3, 64, [1, 2, 3], True, tie=True)
  # Saturated: uniform confidences -> every pixel and every fragment kept
  # (maximum output size h*w*F per object).
  corresp_case(corresp, 'o3_full_s5', 5, 6, 8, 3, 64, [1, 2, 3], True,
               sharp=0.0)
  # No pixel above tau_a for any object -> empty dict (corresp.py:49).
  corresp_case(corresp, 'o3_empty_s7', 7, 6, 8, 3, 64, [1, 2, 3], True,
               sharp=0.0, tau_a=0.9)
  # Ragged size (not a multiple of 64 pixels) and a different scale.
  corresp_case(corresp, 'o2_ragged_s6', 6, 7, 13, 2, 64, [1, 2], True,
               output_scale=0.5, tau_a=0.3, tau_b=0.8)
  fragment_case(fragment, 'ellipsoid_s0', 0, 1500, 64)
  fragment_case(fragment, 'ellipsoid_s1', 1, 400, 16)


if __name__ == '__main__':
  main()
