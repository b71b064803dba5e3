#!/usr/bin/env python
"""EPOS inference benchmark on MI355X: images/sec end-to-end (CNN + correspondence
extraction + PnP-RANSAC), BASELINE.json's metric, on synthetic 640x480 frames.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch with the image tensor already
resident in HBM and the pose records on the host at the end of the step. Default
batch: N = 1 -> 1 image = BASELINE config C2 ("YCB-V, xc65, batch=1 on 1 MI355X": 21
objects, 64 fragments, 5 target objects per image); N > 1 -> 4 images per GPU = the
per-GPU shard of config C3 ("batch=32 sharded across 8 MI355X"; weak scaling: the
shard stays 4 images at N = 2 and 4 too). The ranks exchange nothing per step; their
pose records are gathered once (RCCL all_gather) before the timed region closes. Weights are random-init with the
reference's initialisers (no network access for the released checkpoints).

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  "roofline":     matrix-pipe roofline of the dominant kernel (pointwise_gemm_split_f32:
                  fp32 GEMM as six bf16 piece products), measured with HIP events
                  around every GEMM launch; its "traffic" (HBM-side bytes per launch) is
                  measured in the same run by two rocprofv3 --pmc child runs after the
                  timed region (--traffic static|off to skip),
  "cpu_baseline": the CPU oracle (torch-CPU net + numpy corresp + C RANSAC) timed
                  on this host on a bounded sample (N=1, rank 0 only).
"""
import argparse
import math
import ctypes
import json
import os
import sys
import time

# One hardware queue per pipeline stream (+ the default stream): with the runtime's
# default of 4, the fourth pipeline shares a queue with another one and the two
# serialise against each other (measured: depth 4 = 208 images/s with 4 queues, 232
# with >= 5). Must be set before the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np     # noqa: E402
import torch           # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from epos_amd import dist as edist          # noqa: E402
from epos_amd import pipeline, synthetic, weights   # noqa: E402
from epos_amd import _lib                   # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 roof
BF16_MFMA_PEAK_TFLOPS = 2516.6  # dense v_mfma_f32_32x32x16_bf16: 1024 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
# The split kernel spends six bf16 piece products per fp32 product, so its roof in
# ALGORITHMIC (fp32) flops is the bf16 peak / 6; the fp16-pair kernel (round 3, default)
# spends three fp16 products (same MFMA rate as bf16): peak / 3.
SPLIT_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
H2_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0


def gemm_family():
  """Which kernel the plan's GEMMs run on (the library's environment switches)."""
  if os.environ.get('EPOS_GEMM_SPLIT', '1') == '0':
    return 'fp32'
  if os.environ.get('EPOS_GEMM_H2', '1') == '0':
    return 'split'
  return 'h2' 
ALGO_GFLOP_C2 = 455.0           # SURVEY.md App. A, per image

# EPOS_GEMM_SPLIT=0 (the fp32-MFMA comparison line of BASELINE.md's table): those kernels are
# test-only since round 6 -- the line is then measured on the test build of the library
if gemm_family() == 'fp32' and not os.environ.get('EPOS_HIP_LIB'):
  from epos_amd import build as _hip_build
  os.environ['EPOS_HIP_LIB'] = _hip_build.REF_LIB_PATH


def workload_name(args, B):
  """BASELINE.json's config the flags describe (C2 is the metric's; the others are the
  parity-test configurations, benched for BASELINE.md's table)."""
  if args.model_variant == 'resnet_v1_101_beta':
    return 'C5 (LM-O-shaped, ResNet-v1-101-beta, %d objects, batch %d)' % (args.num_objs, B)
  if (args.height, args.width) == (540, 720) and args.num_objs == 30:
    return 'C4 (T-LESS-shaped 720x540, 30 objects, %d instances per target object)' % args.instances
  if (args.height, args.width) == (480, 640) and args.num_objs == 1:
    return 'C1 (one object) on the GPU path'
  if (args.height, args.width) == (480, 640) and args.num_objs == 21:
    return ('C2' if B == 1 else 'C3 per-GPU shard (batch %d per GPU)' % B if B == 4
            else 'C2-shaped, batch %d per GPU' % B)
  return 'custom, batch %d per GPU' % B


def planted_suffix(args):
  return (' planted (known poses rendered into the heads of the target objects, %d %% outlier '
          'pixels, 1 px noise)' % round(100 * args.planted_outliers)) if args.planted_poses else ''


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--timed-repeats', type=int, default=5,
                  help='the timed region of --steps steps is run this many times back to back, '
                       'each bracketed by barrier + synchronize on both sides; value = all '
                       'images / all region time (a 20-step region is ~50 ms: one region '
                       'cannot resolve a 1 %% change), the regions\' spread is reported')
  ap.add_argument('--batch-per-gpu', type=int, default=None,
                  help='images per GPU per step; default 1 at --gpus 1 (config C2), 4 '
                       'at --gpus > 1 (the per-GPU shard of config C3: 32 images / 8)')
  ap.add_argument('--height', type=int, default=480)
  ap.add_argument('--width', type=int, default=640)
  ap.add_argument('--num-objs', type=int, default=21)
  ap.add_argument('--num-frags', type=int, default=64)
  ap.add_argument('--objs-per-image', type=int, default=5)
  ap.add_argument('--instances', type=int, default=1,
                  help='instances requested per target object (1 = C2/C3/C5: GC-RANSAC only; '
                       '> 1 = the multi-instance search of config C4, T-LESS-like)')
  ap.add_argument('--model-variant', default='xception_65',
                  choices=['xception_65', 'resnet_v1_101_beta'],
                  help='backbone (the headline workload C2 is xception_65; '
                       'resnet_v1_101_beta with --num-objs 15 --batch-per-gpu 8 is C5)')
  ap.add_argument('--weights', default='init', choices=['init', 'heavy-tailed'],
                  help='"init": the reference\'s initialisers; "heavy-tailed": the same checkpoint '
                       'with the tails of a trained network in every GEMM matrix (1 %% of the '
                       'weights scaled by 1e-10, every fourth column log-normal sigma 3: '
                       'weights.heavy_tailed) -- same shapes, same flops; it must run on the same '
                       'kernels at the same rate (config.h2_layers / h2_refused)')
  ap.add_argument('--planted-poses', action='store_true',
                  help='the fitting stage gets EPOS-like work: after the network ran, the head '
                       'values of every target object are overwritten (three scatter launches '
                       'per image, inside the timed step) by a rendering of the object at a '
                       'KNOWN pose -- object confidence, two live fragments per pixel, '
                       'fragment-local 3D coordinates with 1 px-equivalent noise, and '
                       '--planted-outliers of the masked pixels replaced by outliers '
                       '(synthetic.planted_scene). The CNN does its full work; correspondences '
                       'and PnP-RANSAC see thousands of inliers per object instead of ~30, and '
                       'the recovered poses are checked against the planted ones')
  ap.add_argument('--planted-outliers', type=float, default=0.5,
                  help='fraction of outlier pixels per planted object (0.3 / 0.5 / 0.7)')
  ap.add_argument('--no-calibrate', action='store_true',
                  help='keep the raw random-init logits layers (every confidence '
                       'then stays below tau_a and corr/RANSAC get no work)')
  ap.add_argument('--no-graph', action='store_true')
  ap.add_argument('--launch-queue', type=int, default=int(os.environ.get('EPOS_LAUNCH_QUEUE', '2')),
                  help='batches enqueued per pipeline before the oldest is collected '
                       '(EposPipeline(queue=)): with 2 a pipeline\'s next batch is already in its '
                       'stream when the current one finishes, so the stream does not wait for the '
                       'host between two batches (default 2: +0.3..+0.8 %, profiles/r06/ab_launch_queue_*.txt; '
                       'the strictly serial figure serial_depth1 always launches and collects one at a time)')
  ap.add_argument('--pipeline-depth', type=int, default=0,
                  help='batches in flight per GPU: with >= 2, the fitting tail of '
                       'step i overlaps the network of step i+1 (two independent '
                       'plans on two HIP streams); 1 = strictly serial steps. 0 (default) '
                       '= 4 at one image per batch, 2 from four images per batch on (a batch '
                       'of four fills the chip by itself: 432 vs 421 images/s at depth 2 vs 4, '
                       'same box), 3 in between')
  ap.add_argument('--sparse-heads', action='store_true',
                  help='evaluate the fragment heads only for the target objects of '
                       'each image (identical poses, fewer FLOPs); default: dense '
                       'heads as model.predict defines them')
  ap.add_argument('--fitting-method', default='progressive_x',
                  choices=['progressive_x', 'opencv_ransac'],
                  help='infer.py --fitting_method; the metric is quoted on the default')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-roofline', action='store_true')
  ap.add_argument('--cpu-baseline-images', type=int, default=8,
                  help='frames of the same workload the CPU oracle is timed on '
                       '(~1.6 s each on 10 threads: a bounded 10-30 s sample)')
  ap.add_argument('--traffic', default='live', choices=['live', 'static', 'off'],
                  help='roofline.traffic: "live" = two rocprofv3 --pmc passes (FETCH_SIZE, '
                       'WRITE_SIZE; separate runs, no trace options) over a short child run '
                       'of this script, launched after the timed region (falls back to '
                       '"static" if rocprofv3 is missing or fails); "static" = the committed '
                       'summary of the last collection under profiles/; "off" = null')
  ap.add_argument('--no-stage-times', action='store_true',
                  help='skip the serial (depth 1) steps with per-stage HIP events that '
                       'follow the timed region')
  return ap.parse_args()


def cpu_baseline(ckpt, store, args, frames):
  """Oracle chain on the host cores: the reference's CPU path cannot run (no TF,
  no progressive-x), so this is the build's restatement ("port")."""
  from oracle import corresp_ref, net_ref, pnp_ref
  threads = min(10, os.cpu_count() or 1)      # infer.py:693-698: 10 TF threads
  torch.set_num_threads(threads)
  n = 0
  t0 = time.time()
  stage = {'prediction': 0.0, 'establish_corr': 0.0, 'fitting': 0.0}
  for i in range(frames):
    img = synthetic.image(i, args.height, args.width)[None]
    tgt = synthetic.targets(i, args.num_objs, args.objs_per_image)
    t = time.time()
    pred = net_ref.predict(img, ckpt, num_objs=args.num_objs,
                           num_frags=args.num_frags)
    stage['prediction'] += time.time() - t
    t = time.time()
    corr = corresp_ref.establish_many_to_many(
        pred['pred_obj_conf'][0], pred['pred_frag_conf'][0],
        pred['pred_frag_loc'][0], list(tgt), store.dp_model['obj_ids'],
        store.frag_centers, store.frag_sizes, 0.25, 0.1, 0.5, True)
    stage['establish_corr'] += time.time() - t
    t = time.time()
    for obj_id, c in corr.items():              # serial per object, infer.py:412
      if len(c['coord_2d']) < 6:
        continue
      pnp_ref.find6DPoses(c['coord_2d'], c['coord_3d'], synthetic.YCBV_K,
                          params=pnp_ref.default_params(max_model_number=1),
                          seed=obj_id)
    stage['fitting'] += time.time() - t
    n += 1
  dt = time.time() - t0
  return {
      'value': n / dt, 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
      'sample': '%d synthetic %dx%d frame(s), %d objs/%d targets: torch-CPU fp32 '
                'net (%d threads) + numpy corresp + single-thread C PnP-RANSAC '
                '(oracle/); stage seconds %s' % (
                    n, args.width, args.height, args.num_objs,
                    args.objs_per_image, threads,
                    {k: round(v, 3) for k, v in stage.items()}),
  }


def measure_traffic_live(args, timeout=240):
  """HBM-side bytes per GEMM launch, measured NOW: two separate `rocprofv3 --pmc` passes
  (FETCH_SIZE, then WRITE_SIZE; no trace options -- MI355X_MICROARCH.md, HBM / rocprofv3)
  over a child run of this script (1 plan, 4 steps of the same workload, graph replay), the
  per-dispatch counter rows of the GEMM kernels averaged. gfx950: FETCH_SIZE tallies the
  128-B requests of wide coalesced reads at 64 B -> doubled; KB -> bytes. Returns
  (bytes per launch, description) or (None, reason)."""
  import csv
  import glob
  import shutil
  import subprocess
  import tempfile
  rp = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
  if not os.path.exists(rp):
    return None, 'rocprofv3 not found'
  # Never nest: when this process itself runs under a profiler (rocprofv3 exports its
  # options and preloads its tool library), a child would inherit the tracing set-up and
  # combine it with --pmc -- the combination that must not be run.
  prof_keys = ('ROCPROF', 'ROCP_', 'ROCTRACER', 'HSA_TOOLS_LIB', 'ROCTX')
  if any(k.startswith(prof_keys) for k in os.environ) or \
      'rocprof' in os.environ.get('LD_PRELOAD', '').lower():
    return None, 'this process runs under a profiler'
  avg, launches = {}, {}
  for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    d = tempfile.mkdtemp(prefix='epos_pmc_', dir='/tmp')
    cmd = [rp, '--pmc', counter, '--output-format', 'csv', '-d', d, '--', sys.executable,
           os.path.abspath(__file__), '--gpus', '1', '--steps', '4', '--warmup', '1',
           '--timed-repeats', '1', '--pipeline-depth', '1', '--batch-per-gpu', str(args.batch_per_gpu),
           '--height', str(args.height), '--width', str(args.width),
           '--num-objs', str(args.num_objs), '--num-frags', str(args.num_frags),
           '--objs-per-image', str(args.objs_per_image), '--model-variant',
           args.model_variant, '--no-cpu-baseline', '--no-roofline', '--no-stage-times',
           '--traffic', 'off', '--weights', args.weights]
    if args.sparse_heads:
      cmd.append('--sparse-heads')
    env = dict(os.environ, TMPDIR='/tmp', EPOS_BENCH_CHILD='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
      env.pop(k, None)
    try:
      subprocess.run(cmd, cwd='/tmp', env=env, timeout=timeout, check=True,
                     stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
      tot, n = 0.0, 0
      for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
          for r in csv.DictReader(fh):
            k = r.get('Kernel_Name', '')
            if r.get('Counter_Name') == counter and ('pointwise_gemm' in k or
                                                     'pointwise_gemv' in k):
              tot += float(r['Counter_Value'])
              n += 1
      if n == 0:
        return None, 'no %s rows for the GEMM kernels' % counter
      avg[counter], launches[counter] = tot / n, n
    except Exception as e:                       # timeout, non-zero exit, parse error
      return None, '%s pass failed: %s' % (counter, type(e).__name__)
    finally:
      shutil.rmtree(d, ignore_errors=True)
  bytes_per_launch = (2.0 * avg['FETCH_SIZE'] + avg['WRITE_SIZE']) * 1024.0
  return bytes_per_launch, (
      'measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate '
      'child runs (4 steps + warm-up of the same workload on one plan, %d / %d GEMM '
      'dispatches), (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch -- gfx950 tallies the '
      '128-B requests of wide reads at 64 B; fabric-side counters, Infinity-Cache hits '
      'included' % (launches['FETCH_SIZE'], launches['WRITE_SIZE']))


def depthwise_roofline(pipe, steps):
  """The second kernel family of the step, HBM-bound: HIP events around every depthwise
  launch of the plan (same eager passes as gemm_roofline), algorithmic bytes = input read
  once + output written once."""
  net = pipe.net
  s = net._stream()
  evs = []
  for _ in range(steps):
    for name, fn in net.ops:
      if net.op_kind.get(name) == 'dw':
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(s)
        e1.record()
        evs.append((name, e0, e1))
      else:
        fn(s)
  torch.cuda.synchronize()
  ms = sum(e0.elapsed_time(e1) for _, e0, e1 in evs)
  nbytes = sum(net.op_bytes.get(n, 0) for n, _, _ in evs)
  if not evs or ms <= 0:
    return None
  gbs = nbytes / (ms * 1e-3) / 1e9
  return {'kernel': 'depthwise3x3_s1_kernel / depthwise3x3_kernel', 'bound': 'hbm',
          'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s',
          'frac': round(gbs / 8000.0, 4),
          'launches_per_image': len(evs) // steps // net.B,
          'avg_launch_us': round(ms * 1e3 / len(evs), 2),
          'algorithmic_bytes_per_launch': round(nbytes / len(evs)),
          'note': 'HIP events around 8-16 us kernels include ~4 us of launch gap per launch '
                  '(rocprofv3 kernel trace of the same plan: 9.5-11 us, 15-16.5 for the '
                  '2048-channel tensors -> ~3.2 TB/s = 0.40, profiles/r03/rocprofv3_kernel_'
                  'stats_depth1.csv); tensors of 4-60 MB, mostly served by the 256 MB '
                  'Infinity Cache; a plain device copy of the same tensors runs at '
                  '6.4-7.0 TB/s (profiles/r02/depthwise_threads_ab.txt)'}


def gemm_roofline(pipe, steps):
  """HIP-event timing of every pointwise-GEMM launch of the plan (events on
  the stream the kernels are launched on), averaged over `steps` passes."""
  net = pipe.net
  s = net._stream()
  names = [n for n, _ in net.ops]
  is_gemm = [net.op_kind.get(n) == 'gemm' for n in names]
  evs = []
  for _ in range(steps):
    row = []
    for (name, fn), g in zip(net.ops, is_gemm):
      if g:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(s)
        e1.record()
        row.append((name, e0, e1))
      else:
        fn(s)
    evs.append(row)
  torch.cuda.synchronize()
  total_ms, launches, flops, abytes = 0.0, 0, 0, 0
  per = {}
  for row in evs:
    for name, e0, e1 in row:
      ms = e0.elapsed_time(e1)
      total_ms += ms
      launches += 1
      flops += net.op_flops[name]
      abytes += net.op_bytes.get(name, 0)
      per.setdefault(name, []).append(ms)
  achieved = flops / (total_ms * 1e-3) / 1e12
  fam = gemm_family()
  if fam == 'h2':
    kernel, peak = 'pointwise_gemm_h2_f32', H2_PEAK_TFLOPS
    peak_note = ('algorithmic fp32 flops against the fp16 dense MFMA peak (2516.6, the bf16 '
                 'figure: v_mfma_f32_32x32x16_f16 runs at the same rate) / 3: the kernel '
                 'forms every fp32 product from THREE fp16 piece products of two-piece '
                 'round-to-nearest operand splits under power-of-two scales (fp32 in, fp32 '
                 'out, measured error below the fp32 MFMA kernel\'s); frac is therefore '
                 'also the utilisation of the matrix pipe. The bf16 x 6 kernel\'s roof for '
                 'the same work is 419.4, the fp32-MFMA roof 157.3; what bare MFMAs sustain '
                 'at the 1400 W cap is ~2000 (-> ~667 here)')
  elif fam == 'split':
    kernel, peak = 'pointwise_gemm_split_f32', SPLIT_PEAK_TFLOPS
    peak_note = ('algorithmic fp32 flops against the bf16 dense MFMA peak (2516.6) / 6: the '
                 'kernel forms every fp32 product from six bf16 piece products of exact '
                 'three-way operand splits (fp32-equivalent: fp32 in, fp32 out, error below '
                 'the fp32 MFMA kernel\'s); the fp32-MFMA roof of the same work is 157.3')
  else:
    kernel, peak = 'pointwise_gemm_dma_f32', FP32_MFMA_PEAK_TFLOPS
    peak_note = 'dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)'
  # HBM-side bytes per launch: main() replaces this with a measurement of THIS run
  # (measure_traffic_live: two rocprofv3 --pmc child runs); the committed summary of the
  # last collection is the fallback when rocprofv3 is unavailable (--traffic static).
  traffic, traffic_src = None, None
  for rnd in ('r05', 'r04', 'r03', 'r02', 'r01'):
    tpath = os.path.join(ROOT, 'profiles', rnd, 'gemm_hbm_traffic_pmc.json')
    if os.path.exists(tpath):
      with open(tpath) as f:
        traffic = round(json.load(f)['traffic_bytes_per_launch'])
      traffic_src = ('STATIC, not measured in this run: profiles/%s/gemm_hbm_traffic_'
                     'pmc.json (rocprofv3 --pmc passes of tools/profile_round.sh on the '
                     'same plan; PMC counters cannot be read from inside the process)'
                     % rnd)
      break
  return {
      'bound': 'mfma', 'achieved': round(achieved, 2),
      'peak': round(peak, 1), 'unit': 'TFLOP/s',
      'frac': round(achieved / peak, 4), 'traffic': traffic,
      'traffic_unit': 'bytes/launch', 'traffic_source': traffic_src,
      'traffic_measured_in_run': False,   # main() replaces the three fields when --traffic live works
      'algorithmic_bytes_per_launch': round(abytes / max(launches, 1)),
      'kernel': kernel,
      'peak_note': peak_note,
      'frac_of_fp32_mfma_peak': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
      'mfma_pipe_utilisation': round(achieved * {'h2': 3.0, 'split': 6.0}.get(fam, 16.0) /
                                     BF16_MFMA_PEAK_TFLOPS, 4),
      'launches_per_image': launches // steps // net.B,
      'avg_launch_us': round(total_ms * 1e3 / launches, 2),
      'gflop_per_image': round(flops / steps / net.B / 1e9, 1),
      'how': 'sum of algorithmic 2*M*N*K over all launches / sum of HIP-event '
             'durations, %d eager passes after the timed region' % steps,
  }, per


SMI = '/opt/rocm/bin/rocm-smi'


def smi_device_index(dev_index):
  """rocm-smi's index of the HIP device that is benchmarked (HIP_VISIBLE_DEVICES may renumber
  the devices): matched by PCI bus id; None when rocm-smi is missing or nothing matches."""
  import re
  import subprocess
  if not os.path.exists(SMI):
    return None
  try:
    pr = torch.cuda.get_device_properties(dev_index)
    want = '%04x:%02x:%02x' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, pr.pci_device_id)
    out = subprocess.run([SMI, '--showbus'], capture_output=True, text=True, timeout=10).stdout
    rows = re.findall(r'GPU\[(\d+)\]\s*:\s*PCI Bus:\s*([0-9a-fA-F:.]+)', out)
    for idx, bus in rows:
      if bus.lower().startswith(want):
        return int(idx)
    return int(rows[0][0]) if len(rows) == 1 else None
  except Exception:
    return None


def smi_sample(idx):
  """(socket power in W, sclk in MHz) of rocm-smi device `idx`, or (None, None)."""
  import re
  import subprocess
  try:
    o = subprocess.run([SMI, '-d', str(idx), '--showpower', '--showclocks'],
                       capture_output=True, text=True, timeout=5).stdout
    m_ = re.search(r'Power \(W\): ([0-9.]+)', o)
    c_ = re.search(r'sclk clock level[^(]*\(([0-9.]+)Mhz\)', o)
    return (float(m_.group(1)) if m_ else None), (float(c_.group(1)) if c_ else None)
  except Exception:
    return None, None


def main():
  args = parse_args()
  rank, world, local_rank = edist.init_from_env()
  if world != args.gpus and world > 1:
    raise SystemExit('WORLD_SIZE (%d) != --gpus (%d)' % (world, args.gpus))
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X (no CPU fallback); the CPU oracle '
                     'is only timed beside the GPU path.')
  # EPOS_FORCE_DEVICE=0 maps every rank onto one GPU (functional test of the
  # multi-rank flow on a one-GPU box, together with EPOS_DIST_BACKEND=gloo)
  dev_index = int(os.environ.get('EPOS_FORCE_DEVICE', local_rank))
  torch.cuda.set_device(dev_index)
  dev = 'cuda:%d' % dev_index
  # the box's idle power, before this process has put any work on the device
  smi_index = smi_device_index(dev_index) if (world == 1 and not args.no_roofline) else None
  idle_w = None
  if smi_index is not None:
    idle = [smi_sample(smi_index)[0] for _ in range(3)]
    idle = [w for w in idle if w is not None]
    idle_w = round(float(np.mean(idle)), 0) if idle else None
  if args.batch_per_gpu is None:
    args.batch_per_gpu = 1 if world == 1 else 4
  B = args.batch_per_gpu
  # Random-init weights with the reference's initialisers; BatchNorm statistics
  # are randomised so that activations keep an O(0.1..1) scale through the 65
  # layers (identity BN lets them decay to 1e-3, which both starves the heads and
  # flatters the clocks), and the logits layers are calibrated on one frame so
  # that the correspondence / RANSAC stages see YCB-V-like amounts of work.
  ckpt = weights.random_init(args.model_variant, num_objs=args.num_objs,
                             num_frags=args.num_frags, seed=0, randomize_bn=True)
  if args.weights == 'heavy-tailed':
    ckpt = weights.heavy_tailed(ckpt, seed=0)
  from epos_amd import model
  mo = model.ModelOptions(
      model.get_outputs_to_num_channels(args.num_objs, args.num_frags),
      model_variant=args.model_variant)
  store = synthetic.ModelStore(args.num_objs, args.num_frags, seed=0)
  if not args.no_calibrate:
    net0 = model.get_net(ckpt, 1, args.height, args.width, args.num_objs,
                         args.num_frags, mo, device=dev)
    net0.forward(torch.from_numpy(
        synthetic.image(0, args.height, args.width)[None]).to(dev))
    torch.cuda.synchronize()
    synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
    model._NETS.clear()
    del net0
  depth = args.pipeline_depth if args.pipeline_depth > 0 else (4 if B == 1 else 2 if B >= 4 else 3)
  pipes = [pipeline.EposPipeline(
      ckpt, B, args.height, args.width, args.num_objs, args.num_frags, store,
      capacity=1 << 20, max_instances=max(1, args.instances), device=dev,
      use_graph=not args.no_graph, instance=j, sparse_heads=args.sparse_heads,
      model_options=mo, fitting_method=args.fitting_method, queue=max(1, args.launch_queue) if depth > 1 else 1)
           for j in range(depth)]
  pipe = pipes[0]
  # Synthetic frames, resident in HBM before the timed region.
  # a pool size co-prime with the pipeline depth, so that a plan does not see the same frame
  # at every step (plan j takes step i = j, j + depth, ...: frames i % n_pool)
  n_pool = 5
  while math.gcd(n_pool, depth) != 1:
    n_pool += 1
  pool = []
  for j in range(n_pool):
    idx = [rank * 100000 + j * B + b for b in range(B)]
    imgs = np.stack([synthetic.image(i, args.height, args.width) for i in idx])
    tg = [{o: args.instances for o in
           synthetic.targets(i, args.num_objs, args.objs_per_image)} for i in idx]
    pool.append((torch.from_numpy(imgs).to(dev), tg, idx))
  Ks = np.tile(synthetic.YCBV_K, (B, 1, 1))
  # --planted-poses: per pool entry, the rendered head values of its target objects (device
  # resident like the frames) and the poses they were rendered from
  plants = None
  if args.planted_poses:
    if args.sparse_heads:
      raise SystemExit('--planted-poses writes the dense head buffers; not with --sparse-heads')
    plants = []
    for imgs_, tg, idx in pool:
      scenes = [synthetic.planted_scene(i, store, tg[b], synthetic.YCBV_K, pipe.net.out_h,
                                        pipe.net.out_w, args.num_objs, args.num_frags,
                                        outlier_frac=args.planted_outliers, image_in_batch=b)
                for b, i in enumerate(idx)]
      dv = {}
      for key in ('obj', 'frag', 'loc'):
        off = np.concatenate([sc[key][0] for sc in scenes])
        val = np.concatenate([sc[key][1].reshape(len(sc[key][0]), -1) for sc in scenes])
        dv[key] = (torch.from_numpy(off).to(dev), torch.from_numpy(np.ascontiguousarray(val)).to(dev),
                   int(val.shape[1]))
      plants.append({'dev': dv, 'scenes': scenes})

  def planter(j):
    """after_net hook of pool entry j: three scatter launches on the pipeline's stream."""
    if plants is None:
      return None
    dv = plants[j]['dev']

    def plant(p):
      st = ctypes.c_void_p(p.stream.cuda_stream)
      for key, name in (('obj', weights.PRED_OBJ_CONF), ('frag', weights.PRED_FRAG_CONF),
                        ('loc', weights.PRED_FRAG_LOC)):
        off, val, width = dv[key]
        _lib.check(lib.epos_scatter_blocks_f32(
            ctypes.c_void_p(p.net.logits[name].data_ptr()), ctypes.c_void_p(off.data_ptr()),
            ctypes.c_void_p(val.data_ptr()), off.numel(), width, st), 'scatter_blocks')
    return plant
  lib = _lib.load()
  clk = torch.zeros((64, 2), dtype=torch.int64, device=dev)
  clk_stream = None      # created after the timed region (an extra stream changes
                         # the stream -> hardware-queue mapping of the pipelines)

  def run(first, count, probe=False, use=None, one_at_a_time=False):
    """`count` steps; step i is launched on pipes[i % depth] after the step that
    used that pipeline `depth` steps earlier has been collected. Every step is
    complete (its poses on this rank's host) on return; the ranks' records are then
    gathered ONCE (dist.gather_poses: one all_gather over RCCL) -- images are
    independent, so nothing is exchanged per step. Returns the merged pose list."""
    ps = use or pipes
    d = len(ps)
    local, inflight = [], []
    for i in range(first, first + count):
      p = ps[i % d]
      if len(inflight) == d * (1 if one_at_a_time else p.queue):   # the oldest in flight is this pipeline's
        local += inflight.pop(0).collect()[0]
      imgs, tg, idx = pool[i % n_pool]
      p.launch(imgs, Ks, tg, image_ids=idx, seed=i, after_net=planter(i % n_pool))
      inflight.append(p)
      if probe:
        _lib.check(lib.epos_clock_probe(
            ctypes.c_void_p(clk[i % clk.shape[0]].data_ptr()), 200,
            ctypes.c_void_p(clk_stream.cuda_stream)), 'clock_probe')
    while inflight:
      local += inflight.pop(0).collect()[0]
    # rank-independent record capacity: every target object of every image of the
    # `count` steps, two instances each
    return edist.gather_poses(local, max(1, count * B * args.objs_per_image * 2 *
                                         max(1, args.instances)))

  # Set-up, not a step: every plan captures its hipGraph on first use, so each of the
  # `depth` plans is exercised once here (otherwise plans beyond the warm-up count
  # would be captured inside the timed region).
  for j in range(depth):
    imgs, tg, idx = pool[j % n_pool]
    pipes[j].launch(imgs, Ks, tg, image_ids=idx, seed=0, after_net=planter(j % n_pool))
    pipes[j].collect()
  torch.cuda.synchronize()
  def timed_regions(first):
    """R = --timed-repeats regions of EXACTLY --steps steps each, every one bracketed by
    barrier + synchronize on both sides and reduced to its maximum over the ranks.
    Returns (seconds per region, poses of the last region)."""
    secs, poses = [], 0
    for r in range(max(1, args.timed_repeats)):
      torch.cuda.synchronize()
      edist.barrier()
      t0 = time.perf_counter()
      poses = len(run(first + r * args.steps, args.steps))
      torch.cuda.synchronize()
      edist.barrier()
      secs.append(edist.max_over_ranks(time.perf_counter() - t0))
    return secs, poses

  run(0, args.warmup)
  region_s, n_poses = timed_regions(args.warmup)
  n_poses /= max(world, 1)
  repeats = len(region_s)
  elapsed = sum(region_s) / repeats           # seconds per region of --steps steps

  # correspondence statistics of the last step (work actually done by corr/RANSAC)
  totals = pipes[(args.warmup + repeats * args.steps - 1) % depth].last_totals
  images = args.steps * B * world
  value = images / elapsed
  result = {
      'metric': 'images/sec end-to-end (CNN+PnP-RANSAC), YCB-V 640x480',
      'value': round(value, 3), 'unit': 'images/sec', 'n_gpus': world,
      'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': round(elapsed / args.steps * 1e3, 3),
      'timed_repeats': repeats,
      'timed_regions': {
          'images_per_sec_min': round(args.steps * B * world / max(region_s), 2),
          'images_per_sec_max': round(args.steps * B * world / min(region_s), 2),
          'ms_per_step_each': [round(x / args.steps * 1e3, 4) for x in region_s],
          'note': 'value = (timed_repeats x steps x images per step) / (sum of the regions\' '
                  'times); every region is exactly `steps` steps between barrier + '
                  'synchronize pairs, maximum over the ranks'},
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
      'dtype': 'f32', 'data': 'synthetic',
      'config': {
          'workload': workload_name(args, B) + planted_suffix(args) +
                      ': synthetic %dx%d RGB, %s random-init ' % (
                          args.width, args.height, args.model_variant) +
                      '(reference initialisers, randomised BN statistics, logits '
                      'layers %s), %d objects x %d fragments, %d target '
                      'objects/image, batch %d per GPU, dense heads + corr + '
                      'PnP-RANSAC(400 iters, fp64)' % (
                          'raw' if args.no_calibrate else 'calibrated to ~10% '
                          'masked pixels per object', args.num_objs,
                          args.num_frags, args.objs_per_image, B),
          'arithmetic': {
              'h2': 'fp32 activations / weights / accumulators everywhere, as the reference; '
                    '1x1-conv and implicit 3x3 GEMMs: fp32-equivalent products from two-piece '
                    'fp16 splits of both operands (hi = rn(x 2^e), mid = rn((x 2^e - hi) '
                    '2^11): 22-23 significant bits; three of the four piece products, fp32 '
                    'accumulate, the dropped one below 2^-22 of the product and of random '
                    'sign), power-of-two scales per weight column (host) and per activation '
                    'tensor (absmax slots kept by the producing kernels: overflow impossible), '
                    'measured error vs fp64 below the fp32-MFMA kernel at the full C2 size '
                    '(EPOS_GEMM_H2=0: bf16 x 6 kernel, EPOS_GEMM_SPLIT=0: fp32 MFMA kernel)',
              'split': 'fp32 everywhere; GEMMs: fp32-equivalent products from exact three-way '
                       'bf16 splits (six of nine piece products), EPOS_GEMM_H2=0',
              'fp32': 'fp32 everywhere, GEMMs on v_mfma_f32_32x32x2_f32'}[gemm_family()],
          'height': args.height, 'width': args.width,
          'batch_per_gpu': B, 'global_batch': B * world,
          'parallelism': 'dp%d (images sharded, one all_gather of pose records per '
                         'timed region)' % world,
          'dist_backend': (torch.distributed.get_backend()
                           if torch.distributed.is_initialized() else None),
          'rccl_ranks_seen': (torch.distributed.get_world_size()
                              if torch.distributed.is_initialized() else 1),
          'hip_graph': not args.no_graph, 'pipeline_depth': depth,
          'launch_queue': pipe.queue,
          'weights': args.weights,
          # which kernel every GEMM layer of the plan runs on: fp16-pair ("h2") layers and
          # layers whose weight matrix the fp16-pair packer refused (-> bf16 x 6 kernel)
          'h2_layers': len(pipe.net.h2_layers), 'h2_refused': len(pipe.net.h2_refused),
          'h2_refused_names': list(pipe.net.h2_refused)[:8],
          'heads': 'sparse (target objects only)' if args.sparse_heads else 'dense',
          'fitting_method': args.fitting_method,
          'corr_per_slot_last_step': [int(x) for x in totals[:, 1]],
          'poses_per_step': round(n_poses / max(args.steps, 1), 2),
          'algorithmic_gflop_per_image': round(pipe.net.flops / B / 1e9, 1),
      },
  }
  # --planted-poses: the poses that come back are the planted ones (every pool frame once more
  # through pipes[0], outside the timed region): < 1 degree, < 5 mm
  if plants is not None:
    rot, tr, n_gt, n_ok, inl, outside = [], [], 0, 0, [], []
    for j, (imgs, tg, idx) in enumerate(pool):
      est, _ = pipes[0].process_batch(imgs, Ks, tg, image_ids=idx, seed=1000 + j,
                                      after_net=planter(j))
      for b, sc in enumerate(plants[j]['scenes']):
        inl += [2 * (m - o_) for m, o_ in sc['stats'].values()]
        for obj_id, R_gt, t_gt in sc['poses']:
          n_gt += 1
          cand = [synthetic.pose_errors(e['R'], e['t'], R_gt, t_gt) for e in est
                  if e['im_id'] == idx[b] and e['obj_id'] == obj_id]
          if cand:
            best = min(cand, key=lambda c: c[0] + c[1])
            rot.append(best[0]); tr.append(best[1])
            n_ok += int(best[0] < 1.0 and best[1] < 5.0)
          if not cand or not (best[0] < 1.0 and best[1] < 5.0):
            # what the pose outside the bar looked like: a small / far object's translation
            # along the viewing ray is worth a few mm at 1 px of noise; a rotation error near
            # 180 degrees is the other instance's or a symmetric fit
            m_px, o_px = sc['stats'][obj_id]
            outside.append({
                'image': int(idx[b]), 'obj_id': int(obj_id), 'depth_mm': round(float(t_gt[2, 0]), 1),
                'masked_px': int(m_px), 'inlier_correspondences': int(2 * (m_px - o_px)),
                'rot_err_deg': round(best[0], 3) if cand else None,
                'trans_err_mm': round(best[1], 3) if cand else None,
                'trans_err_pct_of_depth': round(100.0 * best[1] / float(t_gt[2, 0]), 3) if cand else None})
    result['planted'] = {
        'outlier_pixel_fraction': args.planted_outliers, 'noise_px': 1.0,
        'planted_poses': n_gt, 'recovered_within_1deg_5mm': n_ok,
        'rot_err_deg_median': round(float(np.median(rot)), 4) if rot else None,
        'rot_err_deg_max': round(float(np.max(rot)), 4) if rot else None,
        'trans_err_mm_median': round(float(np.median(tr)), 4) if tr else None,
        'trans_err_mm_max': round(float(np.max(tr)), 4) if tr else None,
        'inlier_correspondences_per_object_mean': round(float(np.mean(inl)), 1) if inl else 0,
        'ok': bool(n_gt and n_ok == n_gt),
        'outside_the_bar': outside,
        'how': 'after the network ran, epos_scatter_blocks_f32 overwrites pred_obj_conf / '
               'pred_frag_conf / pred_frag_loc of the target objects with a rendering of each '
               'object at a known pose (ray-cast ellipsoid of the synthetic model store; two '
               'live fragments per pixel; 3D noise of 1 px of reprojection; the given fraction '
               'of masked pixels turned into outliers); three launches per step inside the '
               'timed region; the check runs every pool frame once more after it'}
  # Comparability: what a step is made of, at the top level of the line
  fitp = pipe.fit
  result['batch_per_gpu'] = B
  result['pipeline_depth'] = depth
  result['launch_queue'] = pipe.queue
  result['fit'] = {'method': args.fitting_method, 'max_iters': int(fitp.max_iters),
                   'lo_iters': int(fitp.lo_iters), 'gc_sweeps': int(fitp.gc_sweeps),
                   'pearl_iters': int(fitp.pearl_iters), 'max_instances': args.instances,
                   'rounds_per_step': args.instances + (2 if args.instances > 1 else 0)}
  # Decomposition of the TIMED step (same pipelined regime, same steps): the plans' graphs
  # are re-captured without their GEMM / depthwise launches and the same `steps` steps are
  # timed again; step - that = what those kernels cost inside the step, overlap with the
  # other plans in flight included. launches x average <= ms_per_step holds by construction.
  decomp = None
  if not args.no_graph and not args.sparse_heads and not args.no_roofline:
    decomp = {}
    # Without the GEMMs the in-place softmax post-ops are left out as well: the head buffers
    # then keep the probabilities of the last full run, so the correspondence and fitting
    # stages downstream do their REAL work in the decomposition runs (softmax over softmax
    # flattens the scores: no correspondences at 21 objects, all pixels at 1 -- an emptied or
    # overflowing fitting stage would be booked to the GEMMs). The post-ops' own ~0.04 ms are
    # thereby counted with the GEMMs. Without the depthwise launches every GEMM recomputes the
    # same values: everything stays valid.
    for kinds in (('gemm', 'post'), ('dw',)):
      for p_ in pipes:
        with torch.cuda.stream(p_.stream):
          p_.net.capture_alt(kinds)
      run(0, min(args.warmup, 3))
      alt_s, _ = timed_regions(args.warmup)
      decomp[kinds[0]] = sum(alt_s) / len(alt_s) / args.steps * 1e3
    for p_ in pipes:
      p_.net.capture_alt(None)
  # shader clock under the same load: a few extra steps with a spinning probe wave
  # on a side stream (the fp32 MFMA roof is 64 FLOP/clk/SIMD x this clock)
  clk_stream = torch.cuda.Stream(device=dev)
  run(args.warmup + args.steps, min(10, args.steps), probe=True)
  torch.cuda.synchronize()
  clk_h = clk.cpu().numpy()
  clk_h = clk_h[clk_h[:, 1] > 0]
  core_mhz = float((clk_h[:, 0] / clk_h[:, 1]).mean() * 100.0) if len(clk_h) else None
  # Socket power under the same load (rocm-smi, sampled while ~1.2 s of extra pipelined steps
  # run; rank 0 of a one-GPU run only): the step runs close to the 1400 W cap, so the
  # matrix pipe's share of the energy is what is left to win. MFMA-only power at the same
  # clock: profiles/r04/power_by_component_h2.txt.
  power_w = sclk_mhz = None
  if world == 1 and not args.no_roofline:
    import threading
    samples, clocks = [], []
    if smi_index is not None:
      stop = [False]

      def sampler():
        while not stop[0]:
          w, c = smi_sample(smi_index)
          if w is None:
            break
          samples.append(w)
          if c:
            clocks.append(c)
      th = threading.Thread(target=sampler, daemon=True)
      th.start()
      t_end = time.perf_counter() + 1.2
      base = args.warmup + args.steps + 16
      while time.perf_counter() < t_end:
        run(base, 4 * depth)
      stop[0] = True
      th.join(timeout=6)
      if len(samples) > 1:
        power_w = round(float(np.mean(samples[1:])), 0)
      if len(clocks) > 1:
        sclk_mhz = round(float(np.mean(clocks[1:])), 0)
  # Strictly serial steps (ONE plan, depth 1: C2's "batch = 1" read as latency) and the
  # per-stage HIP-event split of such a step (the reference prints the same three
  # stages per image, scripts/infer.py:730-734). Every rank runs them (run() gathers).
  serial = None
  if not args.no_stage_times:
    k = max(3, min(args.steps, 10))
    run(0, 2, use=[pipes[0]], one_at_a_time=True)
    torch.cuda.synchronize()
    edist.barrier()
    t1 = time.perf_counter()
    run(2, k, use=[pipes[0]], one_at_a_time=True)
    torch.cuda.synchronize()
    dt = edist.max_over_ranks(time.perf_counter() - t1)
    stage = {}
    for i in range(k):
      imgs, tg, idx = pool[i % n_pool]
      _, rt = pipes[0].process_batch(imgs, Ks, tg, image_ids=idx, seed=i, timing=True,
                                     after_net=planter(i % n_pool))
      for name, v in rt.items():
        stage[name] = stage.get(name, 0.0) + v * 1e3 / k
    serial = {'images_per_sec': round(k * B * world / dt, 2),
              'ms_per_step': round(dt / k * 1e3, 3), 'steps': k,
              'stage_ms': {n: round(v, 3) for n, v in stage.items()},
              'note': 'pipeline depth 1: one step at a time on one plan; stage_ms = HIP '
                      'events around network / correspondences / PnP-RANSAC of such a '
                      'step (prediction, establish_corr, fitting as scripts/infer.py:'
                      '730-734 prints them)'}
    result['serial_depth1'] = serial
  if rank == 0 and not args.no_roofline:
    roof, _ = gemm_roofline(pipe, max(2, min(args.steps, 5)))
    if args.traffic == 'off':
      roof['traffic'], roof['traffic_source'] = None, 'off'
    elif args.traffic == 'live' and world == 1:
      torch.cuda.synchronize()
      live, how = measure_traffic_live(args)
      if live is not None:
        roof['traffic'], roof['traffic_source'] = round(live), how
        roof['traffic_measured_in_run'] = True
      else:
        roof['traffic_source'] = 'live measurement unavailable (%s); %s' % (
            how, roof['traffic_source'])
    # The HBM view north_star names: images/s x algorithmic bytes per image against the
    # 8 TB/s spec. Bytes per image from the plan itself under the fusion-group rule of
    # SURVEY.md App. A (EposNet.algorithmic_bytes: C2 3.32 GB -- the survey's hand count is
    # 3.30 --, 3.17 at batch 8, C4 4.40, C5 3.64 at batch 8), for every configuration.
    gb = pipe.net.algorithmic_bytes(dense_heads=not args.sparse_heads) / B / 1e9
    roof['hbm_view'] = {
        'bound': 'hbm', 'algorithmic_gb_per_image': round(gb, 3),
        'achieved': round(value / world * gb, 1), 'peak': 8000.0, 'unit': 'GB/s',
        'frac': round(value / world * gb / 8000.0, 4),
        'note': 'whole network, per GPU: the step is bound by the matrix pipe (the '
                'GEMMs are 99 % of the flops at ~150 flop/B), not by HBM'}
    if power_w:
      roof['power_w'] = power_w
      roof['idle_power_w'] = idle_w
      roof['sclk_mhz_rocm_smi'] = sclk_mhz
      need = value / world * pipe.net.flops / B * 3 / 1e12      # fp16 piece products / s
      roof['power_note'] = ('MEASURED: power_w / sclk_mhz_rocm_smi = rocm-smi -d %d (the '
                            'benchmarked device, matched by PCI bus id) sampled while the same '
                            'pipelined steps run, idle_power_w = the same device before this '
                            'process launched anything; cap 1400 W. This step needs %.0f '
                            'fp16-TFLOP/s of piece products.' % (smi_index, need))
      # a MODEL, not a measurement: the constants are profiles/r04/power_by_component_h2.txt's
      # (bare MFMAs: 1490-1670 fp16-TFLOP/s at 1310-1320 W -> ~0.7 pJ per flop above idle)
      roof['power_model_estimate'] = {
          'mfma_share_of_power_above_idle': round(need * 0.7 / max(power_w - (idle_w or 280.0), 1.0), 2),
          'assumes': '0.7 pJ per fp16 flop above idle (profiles/r04/power_by_component_h2.txt)'}
    if core_mhz:
      # sampled in extra steps after the timed region; peak stays the 2.4 GHz figure
      roof['core_clock_mhz_under_load'] = round(core_mhz, 0)
      per_clk = {'pointwise_gemm_split_f32': 1024 / 6.0,
                 'pointwise_gemm_h2_f32': 1024 / 3.0}.get(roof['kernel'], 64)
      roof['peak_at_measured_clock'] = round(per_clk * 1024 * core_mhz * 1e6 / 1e12, 1)
    roof['end_to_end_tflops'] = round(value / world * pipe.net.flops / B / 1e12, 2)
    gemm_gflop = roof['gflop_per_image']
    if args.sparse_heads:     # flops actually executed, not the dense plan's
      net = pipe.net
      dense_heads = sum(f for n, f in net.op_flops.items() if n.startswith('logits/'))
      gemm_gflop = round((sum(f for n, f in net.op_flops.items()
                              if net.op_kind.get(n) == 'gemm') - dense_heads +
                          net._obj_head_flops + pipe.head_flops) / B / 1e9, 1)
      roof['gflop_per_image_sparse_heads'] = gemm_gflop
    roof['achieved_in_pipeline'] = round(
        value / world * gemm_gflop / 1e3, 2)   # GEMM flops only, all streams busy
    if decomp:
      step_ms = result['ms_per_step']
      g_ms = max(step_ms - decomp['gemm'], 0.0)
      d_ms = max(step_ms - decomp['dw'], 0.0)
      nl = roof['launches_per_image'] * B
      roof['in_step'] = {
          'gemm_ms_per_step': round(g_ms, 3), 'depthwise_ms_per_step': round(d_ms, 3),
          'rest_ms_per_step': round(max(step_ms - g_ms - d_ms, 0.0), 3),
          'ms_per_step': step_ms,
          # (a step that is not bound by its GEMMs -- e.g. one object per image, where the
          # serial fitting chain sets the pace -- shows no GEMM cost here: reported as null)
          'avg_launch_us': round(g_ms * 1e3 / max(nl, 1), 2) if g_ms > 0.02 else None,
          'achieved': round(gemm_gflop * B / g_ms, 2) if g_ms > 0.02 else None,
          'frac': round(gemm_gflop * B / g_ms / roof['peak'], 4) if g_ms > 0.02 else None,
          'how': 'the same %d x %d pipelined steps timed again with the GEMM (resp. depthwise) '
                 'launches removed from every plan\'s graph: step - that = the kernels\' '
                 'cost inside the timed regime; launches x avg_launch_us = gemm_ms_per_step '
                 '<= ms_per_step. The GEMM-less graphs also leave out the in-place softmax / '
                 'argmax post-ops (~0.04 ms, counted with the GEMMs here) so that the head '
                 'buffers keep valid probabilities and the correspondence / fitting stages do '
                 'their real work in those runs. roofline.achieved / avg_launch_us above are '
                 'per-launch figures from eager passes (launch gaps included, no overlap) '
                 'and agree with the rocprofv3 kernel trace' % (repeats, args.steps)}
    result['roofline'] = roof
    dwr = depthwise_roofline(pipe, max(2, min(args.steps, 5)))
    if dwr:
      result['roofline_depthwise'] = dwr
  if (rank == 0 and world == 1 and not args.no_cpu_baseline
      and args.fitting_method == 'progressive_x'):
    result['cpu_baseline'] = cpu_baseline(ckpt, store, args,
                                          args.cpu_baseline_images)
  edist.barrier()
  if rank == 0:
    print(json.dumps(result))
  if world > 1:
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
