#!/usr/bin/env python
"""What a user of the drop-in gets: ``infer.py`` end to end on the GPU box, beside
``bench.py`` on the same box (missing #3 / next #2).

Writes a synthetic dataset the way the reference's tools would leave it on disk --
<TF_DATA_PATH>/ycbv_test.tfrecord with N encoded 640 x 480 frames (tf.Example records written
by epos_amd.tfrecord's encoder; 5 annotated target objects per frame out of 21),
<TF_MODELS_PATH>/<model>/{params.yml, fragments.pkl, train/model.npz} -- then runs

    python infer.py --model=<model> --infer_tfrecord_names=ycbv_test [...]

in the configurations below and reports images/s from the first decode to the CSV on disk
(infer.py's own "Throughput:" line) next to bench.py's value, and whether the CSVs agree
(every column but ``time``, which is a measurement).

The checkpoint is bench.py's: random init with the reference's initialisers, logits layers
calibrated on one frame so that corr / fitting see YCB-V-like amounts of work.

    python tools/infer_end_to_end.py --frames 600 --out profiles/r06/infer_end_to_end.txt
"""
import argparse
import io
import json
import os
import pickle
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def natural_like_image(i, h, w):
  """Smooth structure + mild noise: compresses and decodes like a photograph (a pure-noise
  frame is a 450 KB JPEG that takes three times as long to decode)."""
  rng = np.random.RandomState(50000 + i)
  yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
  img = np.zeros((h, w, 3), np.float32)
  for c in range(3):
    for _ in range(6):
      fx, fy = rng.uniform(0.002, 0.05, 2)
      ph = rng.uniform(0, 2 * np.pi)
      img[:, :, c] += rng.uniform(10, 40) * np.sin(fx * xx + fy * yy + ph)
  for _ in range(8):                                    # a few hard-edged blobs
    cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(20, 90)
    m = (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
    img[m] += rng.uniform(-60, 60, 3)
  img += 128 + rng.standard_normal((h, w, 3)) * 6
  return np.clip(img, 0, 255).astype(np.uint8)


def write_dataset(root, n, h, w, num_objs, fmt):
  from PIL import Image
  from epos_amd import synthetic, tfrecord
  data = os.path.join(root, 'data')
  os.makedirs(data, exist_ok=True)
  recs, nbytes, encoded = [], 0, {}
  for i in range(n):
    if i % 64 not in encoded:          # 64 distinct frames, encoded once each
      buf = io.BytesIO()
      Image.fromarray(natural_like_image(i % 64, h, w)).save(
          buf, format=fmt, **({'quality': 90} if fmt == 'JPEG' else {}))
      encoded[i % 64] = buf.getvalue()
    enc = encoded[i % 64]
    nbytes += len(enc)
    tg = synthetic.targets(i, num_objs, 5)
    ids = sorted(tg)
    recs.append(tfrecord.encode_example({
        'image/scene_id': [48 + i // 1000], 'image/im_id': [i],
        'image/path': [('rgb/%06d.%s' % (i, fmt.lower())).encode()],
        'image/encoded': [enc], 'image/height': [h], 'image/width': [w],
        'image/channels': [3],
        'image/camera/fx': [float(synthetic.YCBV_K[0, 0])],
        'image/camera/fy': [float(synthetic.YCBV_K[1, 1])],
        'image/camera/cx': [float(synthetic.YCBV_K[0, 2])],
        'image/camera/cy': [float(synthetic.YCBV_K[1, 2])],
        'image/object/id': ids, 'image/object/visibility': [0.9] * len(ids)}))
  tfrecord.write_records(os.path.join(data, 'ycbv_test.tfrecord'), recs)
  return data, nbytes / n


def write_model(root, name, h, w, num_objs, num_frags, first_frame):
  import torch
  from epos_amd import model, synthetic, weights
  mdir = os.path.join(root, 'models', name)
  os.makedirs(os.path.join(mdir, 'train'), exist_ok=True)
  with open(os.path.join(mdir, 'params.yml'), 'w') as f:
    f.write('dataset: ycbv\nmodel_variant: xception_65\nnum_frags: %d\n'
            'infer_crop_size: "%d,%d"\n' % (num_frags, w, h))
  store = synthetic.ModelStore(num_objs, num_frags, seed=0)
  with open(os.path.join(mdir, 'fragments.pkl'), 'wb') as f:
    pickle.dump({'frag_centers': store.frag_centers, 'frag_sizes': store.frag_sizes}, f)
  ckpt = weights.random_init(num_objs=num_objs, num_frags=num_frags, seed=0,
                             randomize_bn=True)
  net0 = model.get_net(ckpt, 1, h, w, num_objs, num_frags)
  net0.forward(torch.from_numpy(first_frame[None].astype(np.float32)).cuda())
  torch.cuda.synchronize()
  synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
  weights.save_npz(os.path.join(mdir, 'train', 'model.npz'), ckpt)
  return os.path.join(root, 'models')


def run_infer(models, data, name, tag, extra, log):
  cmd = [sys.executable, os.path.join(ROOT, 'infer.py'), '--model=' + name,
         '--infer_tfrecord_names=ycbv_test', '--infer_name', tag] + extra
  t0 = time.time()
  out = subprocess.run(cmd, env=dict(os.environ, TF_MODELS_PATH=models, TF_DATA_PATH=data),
                       capture_output=True, text=True, timeout=3000)
  wall = time.time() - t0
  if out.returncode != 0:
    log('FAILED %s\n%s' % (' '.join(cmd), out.stdout[-3000:] + out.stderr[-3000:]))
    return None
  thr = [l for l in out.stdout.split('\n') if l.startswith('Throughput:')][-1]
  plan = [l for l in out.stdout.split('\n') if l.startswith('plan:')][-1]
  m = re.search(r'= ([0-9.]+) images/s \(inference loop ([0-9.]+) s = ([0-9.]+) images/s', thr)
  stage = [re.findall(r'(prediction|establish_corr|fitting|total time): ([0-9.]+)', l)
           for l in out.stdout.split('\n') if l.startswith('Image:')]
  host = [l for l in out.stdout.split('\n') if l.startswith('Host time per step')]
  mean = {}
  for row in stage[len(stage) // 10:]:
    for k, v in row:
      mean[k] = mean.get(k, 0.0) + float(v) / max(1, len(stage) - len(stage) // 10)
  csv = os.path.join(models, name, 'infer', 'estimated-poses_%s.csv' % tag)
  rows = open(csv).read().strip().split('\n')
  return {'tag': tag, 'extra': ' '.join(extra), 'images_per_s': float(m.group(1)),
          'loop_images_per_s': float(m.group(3)), 'process_wall_s': round(wall, 1), 'plan': plan,
          'mean_stage_ms': {k: round(v * 1e3, 3) for k, v in mean.items()},
          'rows': [','.join(r.split(',')[:-1]) for r in rows], 'n_rows': len(rows) - 1,
          'host': host[-1] if host else '',
          'csv_bytes': os.path.getsize(csv)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--frames', type=int, default=600)
  ap.add_argument('--format', default='JPEG', choices=['JPEG', 'PNG'])
  ap.add_argument('--out', default=None)
  ap.add_argument('--keep', default=None, help='directory to build the dataset in (kept)')
  ap.add_argument('--no-bench', action='store_true')
  ap.add_argument('--threads', default='', help='comma list of --decode_threads values to sweep')
  args = ap.parse_args()
  lines = []

  def log(s=''):
    print(s, flush=True)
    lines.append(s)
  h, w, O, F = 480, 640, 21, 64
  root = args.keep or tempfile.mkdtemp(prefix='epos_e2e_', dir='/tmp')
  t0 = time.time()
  data, avg_bytes = write_dataset(root, args.frames, h, w, O, args.format)
  models = write_model(root, 'ycbv-xc65', h, w, O, F, natural_like_image(0, h, w))
  log('dataset: %d %s frames %dx%d (%.0f KB each on average), 21 objects x 64 fragments, 5 target '
      'objects per frame; written in %.1f s; host cores: %d' % (
          args.frames, args.format, w, h, avg_bytes / 1024, time.time() - t0, os.cpu_count()))
  # decode cost of ONE thread (what the prefetcher's threads have to hide)
  from epos_amd import frames as eframes
  fr = eframes.scan_tfrecords([os.path.join(data, 'ycbv_test.tfrecord')], (w, h), 480, None)
  buf = np.empty((h, w, 3), np.uint8)
  t0 = time.time()
  for f in fr[:50]:
    f.load(buf)
  log('one decoder thread: %.2f ms per frame' % ((time.time() - t0) / 50 * 1e3))
  runs = []
  configs = [('serial_dense', ['--pipeline_depth', '1', '--sparse_heads', 'false']),
             ('default_dense', ['--sparse_heads', 'false']),
             ('default', []),
             ('dense_queue2', ['--sparse_heads', 'false', '--launch_queue', '2']),
             ('default_queue2', ['--launch_queue', '2'])]
  for t in [x for x in args.threads.split(',') if x]:
    configs.append(('dense_threads%s' % t, ['--sparse_heads', 'false', '--decode_threads', t]))
  for tag, extra in configs:
    r = run_infer(models, data, 'ycbv-xc65', tag, extra, log)
    if r is None:
      continue
    runs.append(r)
    log('infer.py %-16s %7.1f images/s first decode -> CSV (loop %7.1f); %d poses; %s; mean '
        'per-image stage ms %s; %s' % (tag, r['images_per_s'], r['loop_images_per_s'],
                                       r['n_rows'], r['plan'], r['mean_stage_ms'], r['host']))
  if runs:
    base = runs[0]
    for r in runs[1:]:
      log('CSV %s vs %s (all columns but time): %s' % (
          r['tag'], base['tag'], 'IDENTICAL' if r['rows'] == base['rows'] else 'DIFFERENT'))
  if not args.no_bench:
    for extra, label in (([], 'dense'), (['--sparse-heads'], 'sparse heads')):
      out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline',
                            '--traffic', 'off', '--no-roofline'] + extra,
                           capture_output=True, text=True, timeout=1200)
      try:
        d = json.loads(out.stdout.strip().split('\n')[-1])
        log('bench.py (%s, same box, frames resident in HBM): %.1f images/s' % (label, d['value']))
        for r in runs:
          if ('sparse' in r['plan']) == bool(extra) and '4 step(s)' in r['plan']:
            log('  infer.py %s / bench.py = %.3f' % (r['tag'], r['images_per_s'] / d['value']))
      except Exception as e:
        log('bench.py failed: %r %s' % (e, out.stderr[-2000:]))
  if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, 'w') as f:
      f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
  main()
