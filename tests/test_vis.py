"""--vis without the renderer (epos_amd/vis.py): grid layout, colourings, the z-buffered
pose overlay and the fragment-field images on synthetic inputs (CPU), and the CLI switch
end to end (GPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_grid_and_colourings():
  from epos_amd import vis
  t = [np.full((225, 300, 3), 10 * (i + 1), np.uint8) for i in range(5)]
  g = vis.build_grid(t, vis.TILE_SIZE)                  # 2 rows x 3 columns (vis.py:64-66)
  assert g.shape == (450, 900, 3)
  assert g[0, 0, 0] == 10 and g[0, 300, 0] == 20 and g[225, 0, 0] == 40
  assert g[225, 600].sum() == 0                          # the empty sixth cell
  with pytest.raises(ValueError):
    vis.build_grid([np.zeros((10, 10, 3), np.uint8)], vis.TILE_SIZE)
  lab = np.array([[0, 1], [2, 257]])
  c = vis.colorize_label_map(lab)
  assert c.shape == (2, 2, 3) and c.dtype == np.uint8 and c[0, 0].sum() == 0
  assert not np.array_equal(c[0, 1], c[1, 0]) and np.array_equal(c[0, 1], c[1, 1])
  with pytest.raises(ValueError):
    vis.colorize_label_map(np.zeros(4, int))
  xyz = np.array([[0., 0, 0], [1, 2, 4]])
  assert vis.colorize_xyz(xyz).tolist() == [[0, 0, 0], [63, 127, 255]]


def test_pose_overlay_and_fragment_fields(tmp_path):
  from epos_amd import synthetic, vis
  store = synthetic.ModelStore(3, 64, seed=0)
  K = synthetic.YCBV_K
  rgb = np.full((480, 640, 3), 100, np.uint8)
  R = np.eye(3)
  near = {'obj_id': 1, 'R': R, 't': np.array([[0.], [0.], [500.]])}
  far = {'obj_id': 2, 'R': R, 't': np.array([[0.], [0.], [900.]])}
  out = vis.overlay_object_poses(rgb, K, [near, far], store)
  assert out.shape == rgb.shape and out.dtype == np.uint8
  # 0.3 * image where nothing is drawn, brighter splats around the principal point
  assert out[5, 5].tolist() == [30, 30, 30]
  centre = out[200:290, 270:360].reshape(-1, 3)
  assert (centre.max(1) > 60).sum() > 50
  # z-buffer: the near object's points win where both project (same centres drawn twice)
  uv1, _ = vis.project(store.frag_centers[1], K, R, near['t'])
  only_far = vis.overlay_object_poses(rgb, K, [far], store)
  assert not np.array_equal(out, only_far)
  # behind the camera: nothing drawn, no exception
  back = {'obj_id': 1, 'R': R, 't': np.array([[0.], [0.], [-500.]])}
  assert np.array_equal(vis.overlay_object_poses(rgb, K, [back], store)[5, 5], [30, 30, 30])
  h, w, O, F = 24, 32, 3, 64
  rng = np.random.RandomState(0)
  pred = {'pred_obj_label': rng.randint(0, O + 1, (h, w)),
          'pred_obj_conf': rng.dirichlet(np.ones(O + 1), (h, w)).astype('f'),
          'pred_frag_conf': rng.dirichlet(np.ones(F), (h, w, O)).astype('f'),
          'pred_frag_loc': rng.standard_normal((h, w, O, F, 3)).astype('f')}
  paths = vis.visualize(rgb, K, pred, [near], 7, store, str(tmp_path), gt_poses=[far],
                        gt_obj_label=pred['pred_obj_label'],
                        flags={'vis_pred_frag_fields': True, 'vis_pred_obj_confs': True})
  names = sorted(os.path.basename(p) for p in paths)
  assert names == ['000007_grid.jpg', '000007_pred_frag_centers.jpg',
                   '000007_pred_frag_coords.jpg', '000007_pred_frag_reconst.jpg']
  from PIL import Image
  g = np.asarray(Image.open(os.path.join(str(tmp_path), '000007_grid.jpg')))
  # input, gt poses, pred poses, gt labels, pred labels + 4 confidences = 9 tiles: 3 x 3
  assert g.shape == (3 * 225, 3 * 300, 3)
  f = np.asarray(Image.open(os.path.join(str(tmp_path), '000007_pred_frag_reconst.jpg')))
  assert f.shape == (h, 3 * w, 3)                       # one row of three object tiles


def test_tfrecord_gt_poses_from_quaternions():
  from epos_amd import tfrecord
  feats = {'image/object/pose/q1': [1.0, 0.0], 'image/object/pose/q2': [0.0, 1.0],
           'image/object/pose/q3': [0.0, 0.0], 'image/object/pose/q4': [0.0, 0.0],
           'image/object/pose/t1': [1.0, 4.0], 'image/object/pose/t2': [2.0, 5.0],
           'image/object/pose/t3': [3.0, 6.0]}
  p = tfrecord._gt_poses(feats, [5, 9], [0, 1])
  assert p[0]['obj_id'] == 5 and np.allclose(p[0]['R'], np.eye(3))
  assert np.allclose(p[1]['R'], np.diag([1.0, -1.0, -1.0]))       # 180 deg about x
  assert p[1]['t'].ravel().tolist() == [4.0, 5.0, 6.0]
  assert tfrecord._gt_poses({}, [5], [0]) is None


@pytest.mark.gpu
def test_infer_vis_end_to_end(tmp_path):
  env = dict(os.environ, TF_MODELS_PATH=str(tmp_path))
  (tmp_path / 'toy').mkdir()
  (tmp_path / 'toy' / 'params.yml').write_text('infer_crop_size: "128,96"\n')
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'infer.py'), '--model=toy', '--synthetic', '2',
       '--num_objs', '3', '--vis', 'true', '--vis_pred_frag_fields', 'true'],
      env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout + out.stderr
  vis_dir = tmp_path / 'toy' / 'vis'                      # scripts/infer.py:577
  names = sorted(os.listdir(str(vis_dir)))
  assert '000000_grid.jpg' in names and '000001_grid.jpg' in names
  assert '000001_pred_frag_reconst.jpg' in names
  assert (tmp_path / 'toy' / 'infer' / 'estimated-poses.csv').exists()
