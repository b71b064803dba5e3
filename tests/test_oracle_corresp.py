"""Pins the corresp / fragment oracles to golden vectors produced by the imported
reference (tests/golden/make_golden.py). CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import corresp_ref, fragment_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
KEYS = ['px_id', 'frag_id', 'coord_2d', 'coord_3d', 'conf', 'conf_obj',
        'conf_frag']


def load_case(path):
  z = np.load(path)
  num_objs = z['frag_centers'].shape[0]
  centers = {o + 1: z['frag_centers'][o] for o in range(num_objs)}
  sizes = {o + 1: z['frag_sizes'][o] for o in range(num_objs)}
  expected = {}
  for oid in z['out_obj_ids']:
    expected[int(oid)] = {k: z['out_%d_%s' % (oid, k)] for k in KEYS}
  return z, num_objs, centers, sizes, expected


CASES = sorted(glob.glob(os.path.join(GOLDEN, 'corresp_*.npz')))


def test_golden_present():
  assert len(CASES) >= 8


@pytest.mark.parametrize('path', CASES, ids=[os.path.basename(p) for p in CASES])
def test_corresp_oracle_matches_reference_golden(path):
  z, num_objs, centers, sizes, expected = load_case(path)
  out = corresp_ref.establish_many_to_many(
      z['obj_confs'], z['frag_confs'], z['frag_coords'],
      gt_obj_ids=list(z['gt_obj_ids']), obj_ids=range(1, num_objs + 1),
      frag_centers=centers, frag_sizes=sizes,
      output_scale=float(z['output_scale']),
      min_obj_conf=float(z['min_obj_conf']),
      min_frag_rel_conf=float(z['min_frag_rel_conf']),
      only_annotated_objs=bool(z['only_annotated']))
  assert sorted(out.keys()) == sorted(expected.keys())
  for oid in expected:
    for k in KEYS:
      a, b = out[oid][k], expected[oid][k]
      assert a.dtype == b.dtype, (oid, k, a.dtype, b.dtype)
      assert a.shape == b.shape, (oid, k)
      assert np.array_equal(a, b), (oid, k)   # bit-exact, ints and floats alike


def test_tie_case_drops_exact_tie():
  z, _, _, _, expected = load_case(os.path.join(GOLDEN, 'corresp_o3_tie_s4.npz'))
  e = expected[1]
  # Pixel (3,5) of the 12x16 map: fragments 7 and 11 kept, tied fragment 9 dropped.
  sel = (e['coord_2d'][:, 0] == 4.0 * 5.5) & (e['coord_2d'][:, 1] == 4.0 * 3.5)
  assert list(e['frag_id'][sel]) == [7, 11]


@pytest.mark.parametrize('name', ['ellipsoid_s0', 'ellipsoid_s1'])
def test_fragment_oracle_matches_reference_golden(name):
  z = np.load(os.path.join(GOLDEN, 'fragment_%s.npz' % name))
  centers, ids = fragment_ref.fragmentation_fps(z['vertices'],
                                                int(z['num_frags']))
  assert np.array_equal(centers, z['frag_centers'])
  assert np.array_equal(ids, z['vertex_frag_ids'])
  sizes = fragment_ref.fragment_sizes(z['vertices'], ids, int(z['num_frags']))
  assert sizes.shape == (int(z['num_frags']),) and (sizes >= 5.0).all()


def test_project_ref_known_answers():
  """oracle/project_ref.py (closest point on a triangle mesh) on a unit square made of
  two triangles: interior, edge, vertex regions and the tie rule."""
  from oracle import project_ref
  verts = np.array([[0., 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])
  faces = np.array([[0, 1, 2], [0, 2, 3]])
  pts = np.array([[0.7, 0.2, 3.0],      # above the first triangle
                  [0.2, 0.7, -2.0],     # below the second
                  [0.5, 0.5, 1.0],      # above the shared diagonal: tie -> face 0
                  [2.0, 0.5, 0.0],      # beyond the edge x = 1
                  [-1.0, -1.0, 0.5],    # beyond the corner (0, 0)
                  [0.5, 2.0, 0.0]])     # beyond the edge y = 1
  out, idx = project_ref.project_pts_to_model(pts, verts, faces)
  exp = np.array([[0.7, 0.2, 0], [0.2, 0.7, 0], [0.5, 0.5, 0], [1.0, 0.5, 0],
                  [0, 0, 0], [0.5, 1.0, 0]])
  np.testing.assert_allclose(out, exp, atol=1e-15)
  assert list(idx) == [0, 1, 0, 0, 0, 1]
