"""PLY reader (epos_amd/ply.py): round trips through its own writer in the three
encodings of the PLY specification, extra vertex properties, polygon fan
triangulation, BOP path layout. No bop_toolkit / model file is available: parity
with inout.load_ply is unpinned (module header)."""
import os

import numpy as np
import pytest

from epos_amd import ply


@pytest.mark.parametrize('fmt', ['ascii', 'binary_little_endian', 'binary_big_endian'])
def test_round_trip(tmp_path, fmt):
  rng = np.random.RandomState(0)
  pts = rng.uniform(-80, 80, (37, 3)).astype(np.float32)
  nrm = rng.standard_normal((37, 3)).astype(np.float32)
  col = rng.randint(0, 256, (37, 3))
  faces = rng.randint(0, 37, (20, 3))
  path = str(tmp_path / 'obj_000001.ply')
  ply.save_ply(path, pts, faces, nrm, col, fmt=fmt)
  m = ply.load_ply(path)
  assert m['pts'].dtype == np.float64
  np.testing.assert_array_equal(m['pts'], pts.astype(np.float64))
  np.testing.assert_array_equal(m['normals'], nrm.astype(np.float64))
  np.testing.assert_array_equal(m['colors'], col.astype(np.float64))
  np.testing.assert_array_equal(m['faces'], faces)


def test_hand_written_ascii_with_quads_comments_and_extra_elements(tmp_path):
  text = '\n'.join([
      'ply', 'format ascii 1.0', 'comment made by hand', 'element vertex 4',
      'property double x', 'property double y', 'property double z',
      'property float texture_u', 'property float texture_v',
      'element face 1', 'property list uchar uint vertex_index',
      'element edge 1', 'property int vertex1', 'property int vertex2', 'end_header',
      '0 0 0 0.0 0.0', '1 0 0 1.0 0.0', '1 1 0 1.0 1.0', '0 1 0.5 0.0 1.0',
      '4 0 1 2 3', '0 1', ''])
  path = str(tmp_path / 'quad.ply')
  with open(path, 'w') as f:
    f.write(text)
  m = ply.load_ply(path)
  assert m['pts'].shape == (4, 3) and m['pts'][3, 2] == 0.5
  np.testing.assert_array_equal(m['faces'], [[0, 1, 2], [0, 2, 3]])
  np.testing.assert_array_equal(m['texture_uv'][2], [1.0, 1.0])


def test_bop_layout_and_errors(tmp_path):
  root = str(tmp_path)
  os.makedirs(os.path.join(root, 'lmo', 'models_eval'))
  for o in ply.BOP_OBJ_IDS['lmo']:
    ply.save_ply(ply.model_path(root, 'lmo', o, 'eval'),
                 np.full((5, 3), float(o)), fmt='ascii')
  models = ply.load_models(root, 'lmo', 'eval')
  assert sorted(models) == [1, 5, 6, 8, 9, 10, 11, 12]
  assert models[9]['pts'][0, 0] == 9.0 and 'faces' not in models[9]
  assert len(ply.BOP_OBJ_IDS['ycbv']) == 21 and len(ply.BOP_OBJ_IDS['tless']) == 30
  bad = str(tmp_path / 'bad.ply')
  with open(bad, 'w') as f:
    f.write('plx\n')
  with pytest.raises(ValueError):
    ply.load_ply(bad)
