"""infer.py (drop-in for scripts/infer.py): flag / params.yml handling on CPU, an
end-to-end synthetic run writing a BOP'19 CSV on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_params_yml_overrides_flag_defaults(tmp_path):
  import infer
  args = infer.build_parser().parse_args(['--model', 'm'])
  assert args.inlier_thresh == 4.0 and args.max_fitting_iterations == 400
  assert args.corr_min_obj_conf == 0.1 and args.corr_min_frag_rel_conf == 0.5
  p = tmp_path / 'params.yml'
  p.write_text('inlier_thresh: 2.5\nnum_frags: 32\ninfer_crop_size: "720,540"\n'
               'unknown_flag: 1\n')
  infer.update_flags(args, str(p))                       # common.py:157-177
  assert args.inlier_thresh == 2.5 and args.num_frags == 32
  assert args.infer_crop_size == '720,540'
  with pytest.raises(ValueError):
    infer.update_flags(args, str(tmp_path / 'params.json.txt'.replace('.txt', ''))
                       if (tmp_path / 'params.json').write_text('{}') or True
                       else None)


def test_every_reference_flag_parses():
  """Every flag the reference's entry point defines (scripts/infer.py:37-146 +
  epos_lib/common.py:56-154; names and defaults in tests/golden/reference_flags.json, written
  by tests/golden/make_flag_names.py from the DEFINE_* calls) is accepted by infer.py's parser,
  with the reference's default wherever the default is a plain number / bool -- including the
  two the reference defines and never reads (--master, --corr_project_to_model)."""
  import json
  import infer
  with open(os.path.join(os.path.dirname(__file__), 'golden', 'reference_flags.json')) as f:
    ref = json.load(f)
  ap = infer.build_parser()
  dests = {a.dest: a for a in ap._actions}
  n = 0
  for src, flags in ref.items():
    for fl in flags:
      assert fl['name'] in dests, '%s: --%s is not accepted' % (src, fl['name'])
      ours = dests[fl['name']].default
      if isinstance(fl['default'], (bool, int, float)) and not dests[fl['name']].required:
        assert ours == fl['default'], (fl['name'], ours, fl['default'])
      n += 1
  assert n == 61
  args = ap.parse_args(['--model', 'm', '--corr_project_to_model', 'True', '--master', 'x'])
  assert args.corr_project_to_model is True and args.master == 'x'


def test_unsupported_model_flags_raise(tmp_path):
  """A params.yml (or flag) that asks for a graph this build does not implement must
  stop the run, not be half-applied (common.py:96-154)."""
  import infer
  ok = infer.build_parser().parse_args(['--model', 'm', '--multi_grid', '1,2,4'])
  infer.check_supported_flags(ok)                        # defaults + multi_grid: fine
  assert infer._as_list(ok.multi_grid, int) == [1, 2, 4]
  for text in ['aspp_with_separable_conv: false\n', 'upsample_logits: true\n',
               'frag_cls_agnostic: true\n', 'image_pyramid: [0.5, 1.0]\n',
               'logits_kernel_size: 3\n', 'decoder_use_separable_conv: false\n',
               'add_image_level_feature: false\n', 'use_bounded_activation: true\n',
               'encoder_output_stride: 16\n', 'model_variant: mobilenet_v2\n',
               'aspp_with_batch_norm: false\n', 'image_pooling_stride: "2,2"\n']:
    args = infer.build_parser().parse_args(['--model', 'm'])
    p = tmp_path / 'params.yml'
    p.write_text(text)
    infer.update_flags(args, str(p))
    with pytest.raises(NotImplementedError):
      infer.check_supported_flags(args)
  # values equal to the defaults, in YAML spellings, pass
  args = infer.build_parser().parse_args(['--model', 'm'])
  (tmp_path / 'params.yml').write_text(
      'image_pyramid: [1.0]\ndecoder_output_stride: [4]\natrous_rates: [12, 24, 36]\n'
      'aspp_with_batch_norm: true\nmulti_grid: [1, 2, 4]\n')
  infer.update_flags(args, str(tmp_path / 'params.yml'))
  infer.check_supported_flags(args)
  assert infer._as_list(args.atrous_rates, int) == [12, 24, 36]


def test_unknown_fitting_method_raises(tmp_path, monkeypatch):
  import infer
  monkeypatch.setenv('TF_MODELS_PATH', str(tmp_path))
  with pytest.raises((ValueError, SystemExit, RuntimeError)):
    infer.main(['--model', 'm', '--fitting_method', 'ceres', '--synthetic', '1'])


def test_fragments_pkl_roundtrip(tmp_path):
  import pickle
  import infer
  centers = {1: np.zeros((64, 3)), 2: np.ones((64, 3))}
  sizes = {1: np.full(64, 5.0), 2: np.full(64, 7.0)}
  with open(tmp_path / 'fragments.pkl', 'wb') as f:
    pickle.dump({'frag_centers': centers, 'frag_sizes': sizes}, f)
  store = infer.load_fragments(str(tmp_path), 64)
  assert store.dp_model['obj_ids'] == [1, 2] and store.frag_sizes[2][0] == 7.0
  with pytest.raises(ValueError):
    infer.load_fragments(str(tmp_path), 32)              # datagen.py:264-268


@pytest.mark.gpu
def test_infer_synthetic_end_to_end(tmp_path):
  env = dict(os.environ, TF_MODELS_PATH=str(tmp_path))
  (tmp_path / 'toy').mkdir()
  (tmp_path / 'toy' / 'params.yml').write_text('infer_crop_size: "128,96"\n')
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'infer.py'), '--model=toy',
       '--synthetic', '3', '--num_objs', '3', '--infer_name', 't'],
      env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout + out.stderr
  csv = tmp_path / 'toy' / 'infer' / 'estimated-poses_t.csv'
  assert csv.exists()
  from epos_amd import bop_io
  res = bop_io.load_bop_results(str(csv))
  for r in res:
    assert r['R'].shape == (3, 3) and r['t'].shape == (3, 1) and r['time'] > 0


@pytest.mark.gpu
def test_infer_operator_path_with_max_correspondences(tmp_path):
  env = dict(os.environ, TF_MODELS_PATH=str(tmp_path))
  (tmp_path / 'toy').mkdir()
  (tmp_path / 'toy' / 'params.yml').write_text('infer_crop_size: "128,96"\n')
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'infer.py'), '--model=toy',
       '--synthetic', '2', '--num_objs', '3', '--max_correspondences', '200',
       '--use_prosac', 'true'],
      env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout + out.stderr
  assert (tmp_path / 'toy' / 'infer' / 'estimated-poses.csv').exists()


@pytest.mark.gpu
def test_infer_opencv_ransac_method_gives_one_unscored_pose_per_object(tmp_path):
  """--fitting_method=opencv_ransac (infer.py:505-528): at most one pose per
  (image, object), score 0.0. PARITY UNPINNED: cv2 is not installable; the model
  comes from this build's P3P-RANSAC."""
  env = dict(os.environ, TF_MODELS_PATH=str(tmp_path))
  (tmp_path / 'toy').mkdir()
  (tmp_path / 'toy' / 'params.yml').write_text('infer_crop_size: "128,96"\n')
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'infer.py'), '--model=toy',
       '--synthetic', '2', '--num_objs', '3', '--fitting_method', 'opencv_ransac'],
      env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout + out.stderr
  rows = (tmp_path / 'toy' / 'infer' / 'estimated-poses.csv').read_text().strip().split('\n')[1:]
  seen = set()
  for r in rows:
    scene, im, obj, score = r.split(',')[:4]
    assert float(score) == 0.0
    assert (scene, im, obj) not in seen
    seen.add((scene, im, obj))


@pytest.mark.gpu
def test_infer_pipeline_depth_gives_the_same_poses(tmp_path):
  """--pipeline_depth 3 (three batches in flight on their own streams) writes the same
  rows as the serial run, up to the time column."""
  rows = {}
  for depth in (1, 3):
    d = tmp_path / ('d%d' % depth)
    (d / 'toy').mkdir(parents=True)
    (d / 'toy' / 'params.yml').write_text('infer_crop_size: "128,96"\n')
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'infer.py'), '--model=toy', '--synthetic', '7',
         '--num_objs', '3', '--pipeline_depth', str(depth)],
        env=dict(os.environ, TF_MODELS_PATH=str(d)), capture_output=True, text=True,
        timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    txt = (d / 'toy' / 'infer' / 'estimated-poses.csv').read_text().strip().split('\n')
    rows[depth] = [','.join(r.split(',')[:-1]) for r in txt]
  assert len(rows[1]) > 1 and rows[1] == rows[3]


@pytest.mark.gpu
def test_infer_from_tfrecord(tmp_path):
  """--infer_tfrecord_names path: records written with the build's own encoder
  (the reference writes them with TensorFlow, scripts/create_tfrecord.py)."""
  import io
  from PIL import Image
  from epos_amd import bop_io, tfrecord
  data = tmp_path / 'data'
  models = tmp_path / 'models'
  data.mkdir(); (models / 'toy').mkdir(parents=True)
  (models / 'toy' / 'params.yml').write_text('infer_crop_size: "128,96"\n')
  rng = np.random.RandomState(0)
  recs = []
  for i in range(2):
    buf = io.BytesIO()
    Image.fromarray(rng.randint(0, 256, (96, 128, 3)).astype(np.uint8)).save(
        buf, format='PNG')
    recs.append(tfrecord.encode_example({
        'image/scene_id': [3], 'image/im_id': [i], 'image/path': [b'x.png'],
        'image/encoded': [buf.getvalue()], 'image/height': [96],
        'image/width': [128], 'image/channels': [3],
        'image/camera/fx': [300.0], 'image/camera/fy': [300.0],
        'image/camera/cx': [64.0], 'image/camera/cy': [48.0],
        'image/object/id': [1, 2, 2], 'image/object/visibility': [0.9, 0.8, 0.7]}))
  tfrecord.write_records(str(data / 'toy_test.tfrecord'), recs)
  env = dict(os.environ, TF_MODELS_PATH=str(models), TF_DATA_PATH=str(data))
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'infer.py'), '--model=toy',
       '--infer_tfrecord_names', 'toy_test', '--synthetic', '1', '--num_objs',
       '3'], env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout + out.stderr
  res = bop_io.load_bop_results(str(models / 'toy' / 'infer' /
                                    'estimated-poses.csv'))
  assert all(r['scene_id'] == 3 and r['obj_id'] in (1, 2) for r in res)


@pytest.mark.gpu
def test_infer_streaming_default_matches_the_serial_dense_run(tmp_path):
  """The drop-in's default configuration since round 6 -- frames decoded by the prefetcher's
  threads into pinned uint8 buffers, uploaded as bytes and cast on the device, four steps in
  flight, fragment heads evaluated for the target objects only -- writes the same CSV rows as
  the strictly serial run with dense heads (--pipeline_depth 1 --sparse_heads false), up to the
  time column; 9 JPEG frames at batch 1 and at batch 2 (padded last batch)."""
  import io
  from PIL import Image
  from epos_amd import tfrecord
  data = tmp_path / 'data'
  data.mkdir()
  rng = np.random.RandomState(0)
  recs = []
  for i in range(9):
    buf = io.BytesIO()
    Image.fromarray(rng.randint(0, 256, (96, 128, 3)).astype(np.uint8)).save(
        buf, format='JPEG', quality=92)
    recs.append(tfrecord.encode_example({
        'image/scene_id': [3], 'image/im_id': [i], 'image/path': [b'x.jpg'],
        'image/encoded': [buf.getvalue()], 'image/height': [96],
        'image/width': [128], 'image/channels': [3],
        'image/camera/fx': [300.0], 'image/camera/fy': [300.0],
        'image/camera/cx': [64.0], 'image/camera/cy': [48.0],
        'image/object/id': [1, 3] if i % 2 else [2], 'image/object/visibility': [0.9] * (1 + i % 2)}))
  tfrecord.write_records(str(data / 'toy_test.tfrecord'), recs)
  rows = {}
  for name, extra in (('serial', ['--pipeline_depth', '1', '--sparse_heads', 'false']),
                      ('default', []), ('batch2', ['--batch', '2'])):
    models = tmp_path / name
    (models / 'toy').mkdir(parents=True)
    (models / 'toy' / 'params.yml').write_text('infer_crop_size: "128,96"\n')
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'infer.py'), '--model=toy',
         '--infer_tfrecord_names', 'toy_test', '--synthetic', '1', '--num_objs', '3'] + extra,
        env=dict(os.environ, TF_MODELS_PATH=str(models), TF_DATA_PATH=str(data)),
        capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    plan = [l for l in out.stdout.split('\n') if l.startswith('plan:')][0]
    assert ('1 step(s) in flight, dense' in plan) == (name == 'serial'), plan
    if name == 'default':
      assert '4 step(s) in flight x 2 enqueued per plan, sparse' in plan, plan
    assert 'Throughput:' in out.stdout
    txt = (models / 'toy' / 'infer' / 'estimated-poses.csv').read_text().strip().split('\n')
    rows[name] = [','.join(r.split(',')[:-1]) for r in txt]
  assert len(rows['serial']) > 1
  assert rows['serial'] == rows['default'] == rows['batch2']


@pytest.mark.gpu
def test_infer_restores_tf_checkpoint(tmp_path):
  """<model>/train/model.ckpt-N.{index,data-*} is restored by variable name
  (infer.py:670-683) through the TensorFlow-free TensorBundle reader."""
  from epos_amd import tf_checkpoint, weights
  models = tmp_path
  (models / 'toy' / 'train').mkdir(parents=True)
  (models / 'toy' / 'params.yml').write_text('infer_crop_size: "128,96"\n')
  ckpt = weights.random_init(num_objs=3, seed=5, randomize_bn=True)
  ckpt['global_step'] = np.asarray(7, np.int64)
  tf_checkpoint.write_checkpoint(str(models / 'toy' / 'train' / 'model.ckpt-7'),
                                 ckpt)
  rng = np.random.RandomState(0)
  np.savez(str(models / 'toy' / 'fragments.npz'), obj_ids=np.arange(1, 4),
           frag_centers=rng.uniform(-50, 50, (3, 64, 3)),
           frag_sizes=rng.uniform(5, 30, (3, 64)))
  env = dict(os.environ, TF_MODELS_PATH=str(models))
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'infer.py'), '--model=toy',
       '--synthetic', '1'], env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout + out.stderr
  assert (models / 'toy' / 'infer' / 'estimated-poses.csv').exists()


@pytest.mark.gpu
def test_fragments_from_bop_ply_models(tmp_path, monkeypatch):
  """No fragments.pkl: infer.py loads <BOP_PATH>/<dataset>/models*/obj_XXXXXX.ply
  (epos_amd/ply.py), fragments the vertices on the GPU and caches fragments.pkl
  (datagen.py:238-296). Centres / sizes must equal the numpy oracle's FPS + size rule
  on the same vertices, and the model-type rule ('eval' models for TUD-L) is honoured."""
  import argparse
  import infer
  from epos_amd import ply
  from oracle import fragment_ref
  bop = tmp_path / 'bop'
  rng = np.random.RandomState(4)
  pts = {}
  for o in ply.BOP_OBJ_IDS['tudl']:
    os.makedirs(os.path.dirname(ply.model_path(str(bop), 'tudl', o, 'eval')),
                exist_ok=True)
    v = rng.uniform(-60, 60, (500 + 37 * o, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True) / (30.0 + 10 * o)
    pts[o] = v.astype(np.float64)
    ply.save_ply(ply.model_path(str(bop), 'tudl', o, 'eval'), v,
                 faces=rng.randint(0, len(v), (50, 3)))
  monkeypatch.setenv('BOP_PATH', str(bop))
  model_dir = tmp_path / 'm'
  model_dir.mkdir()
  args = argparse.Namespace(dataset='tudl', num_frags=64)
  store = infer.fragment_from_bop_models(str(model_dir), args, 'cuda:0')
  assert (model_dir / 'fragments.pkl').exists()
  assert store.dp_model['obj_ids'] == [1, 2, 3]
  for o in store.dp_model['obj_ids']:
    centers, ids = fragment_ref.fragmentation_fps(pts[o], 64)
    np.testing.assert_array_equal(store.frag_centers[o], centers)
    sizes = [max((pts[o][ids == f].max(0) - pts[o][ids == f].min(0)).max(), 5.0)
             for f in range(64)]
    np.testing.assert_array_equal(store.frag_sizes[o], sizes)
  # a dataset without model files falls through (the caller then raises)
  args.dataset = 'ycbv'
  assert infer.fragment_from_bop_models(str(model_dir), args, 'cuda:0') is None


@pytest.mark.gpu
def test_infer_two_ranks_with_an_odd_frame_count(tmp_path):
  """Three frames over two ranks (shards of 2 and 1): the pose gather must not depend on
  the rank-local shard size (two torchrun ranks on the one GPU, gloo for the gather)."""
  import socket
  with socket.socket() as sck:
    sck.bind(('127.0.0.1', 0))
    port = sck.getsockname()[1]
  (tmp_path / 'toy').mkdir()
  (tmp_path / 'toy' / 'params.yml').write_text('infer_crop_size: "128,96"\n')
  env = dict(os.environ, TF_MODELS_PATH=str(tmp_path), EPOS_DIST_BACKEND='gloo',
             EPOS_FORCE_DEVICE='0')
  common = [os.path.join(ROOT, 'infer.py'), '--model=toy', '--synthetic', '3',
            '--num_objs', '3']
  out = subprocess.run(
      [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
       '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
       str(port)] + common + ['--infer_name', 'two'],
      env=env, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
  one = subprocess.run([sys.executable] + common + ['--infer_name', 'one'],
                       env=env, capture_output=True, text=True, timeout=600)
  assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
  from epos_amd import bop_io
  a = bop_io.load_bop_results(str(tmp_path / 'toy' / 'infer' / 'estimated-poses_two.csv'))
  b = bop_io.load_bop_results(str(tmp_path / 'toy' / 'infer' / 'estimated-poses_one.csv'))
  key = lambda r: (r['im_id'], r['obj_id'], r['score'])      # noqa: E731
  assert len(a) == len(b) and len(a) > 0
  for x, y in zip(sorted(a, key=key), sorted(b, key=key)):
    assert (x['im_id'], x['obj_id']) == (y['im_id'], y['obj_id'])
    np.testing.assert_array_equal(x['R'], y['R'])
    np.testing.assert_array_equal(x['t'], y['t'])


@pytest.mark.gpu
def test_infer_eight_ranks_cover_32_frames_exactly_once(tmp_path):
  """The C3 arrangement as a dry run on the one-GPU box: 32 frames over EIGHT torchrun
  ranks (4 per rank, gloo for the gather, every rank on device 0). Every frame must be
  processed by exactly one rank and the merged BOP CSV must equal the single-rank CSV row
  for row (poses bit for bit) -- what tools/run_scale.sh will exercise over RCCL the day an
  8-GPU node runs it."""
  import socket
  with socket.socket() as sck:
    sck.bind(('127.0.0.1', 0))
    port = sck.getsockname()[1]
  (tmp_path / 'toy').mkdir()
  (tmp_path / 'toy' / 'params.yml').write_text('infer_crop_size: "128,96"\n')
  env = dict(os.environ, TF_MODELS_PATH=str(tmp_path), EPOS_DIST_BACKEND='gloo',
             EPOS_FORCE_DEVICE='0')
  common = [os.path.join(ROOT, 'infer.py'), '--model=toy', '--synthetic', '32',
            '--num_objs', '3', '--batch', '4']
  out = subprocess.run(
      [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
       '--nproc-per-node', '8', '--master-addr', '127.0.0.1', '--master-port',
       str(port)] + common + ['--infer_name', 'eight'],
      env=env, capture_output=True, text=True, timeout=1500)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
  one = subprocess.run([sys.executable] + common + ['--infer_name', 'one'],
                       env=env, capture_output=True, text=True, timeout=900)
  assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
  from epos_amd import bop_io
  a = bop_io.load_bop_results(str(tmp_path / 'toy' / 'infer' / 'estimated-poses_eight.csv'))
  b = bop_io.load_bop_results(str(tmp_path / 'toy' / 'infer' / 'estimated-poses_one.csv'))
  key = lambda r: (r['im_id'], r['obj_id'], r['score'])      # noqa: E731
  assert len(a) == len(b) and len(a) > 0
  assert len({r['im_id'] for r in b}) > 16                  # poses all over the 32 frames
  seen = {}
  for x, y in zip(sorted(a, key=key), sorted(b, key=key)):
    assert (x['im_id'], x['obj_id']) == (y['im_id'], y['obj_id'])
    np.testing.assert_array_equal(x['R'], y['R'])
    np.testing.assert_array_equal(x['t'], y['t'])
    seen[(x['im_id'], x['obj_id'], x['score'])] = seen.get((x['im_id'], x['obj_id'], x['score']), 0) + 1
  assert max(seen.values()) == 1                            # no frame fitted twice


def test_entry_points_ask_for_one_hardware_queue_per_pipeline():
  """infer.py and bench.py put GPU_MAX_HW_QUEUES into the environment BEFORE torch (and with
  it the HIP runtime) is imported: with the runtime's default of four queues two of the four
  pipelines share one and serialise against each other (profiles/r06/infer_diag_hw_queues.txt:
  infer.py ran at 0.81 of bench.py until it did this too)."""
  import ast
  for name in ('infer.py', 'bench.py'):
    tree = ast.parse(open(os.path.join(ROOT, name)).read())
    set_at = torch_at = None
    for node in tree.body:
      src = ast.dump(node)
      if set_at is None and 'GPU_MAX_HW_QUEUES' in src and 'setdefault' in src:
        set_at = node.lineno
      if torch_at is None and isinstance(node, (ast.Import, ast.ImportFrom)) and any(
          a.name.split('.')[0] == 'torch' for a in node.names):
        torch_at = node.lineno
    assert set_at is not None and torch_at is not None and set_at < torch_at, (name, set_at, torch_at)
  out = subprocess.run(
      [sys.executable, '-c', 'import os, epos_amd; print(os.environ["GPU_MAX_HW_QUEUES"])'],
      capture_output=True, text=True, cwd=ROOT, env={k: v for k, v in os.environ.items()
                                                   if k != 'GPU_MAX_HW_QUEUES'})
  assert out.stdout.strip() == '8', out.stdout + out.stderr
