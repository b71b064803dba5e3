#!/usr/bin/env python
"""RANSAC-only microbenchmark: the device fitting stage
(epos_find6d_poses_device) on synthetic correspondence sets of N in {500, 2000, 10000}
with 30 / 50 / 70 % outliers (sigma = 1 px), 32 slots per launch sequence (= objects of a
batch), 400 hypotheses each: hypotheses x points per second (the kernels evaluate up to
4 P3P roots per hypothesis against every point), time per slot, pose error vs ground
truth, and the same sets through the single-thread C oracle.

    python tools/bench_ransac.py [--slots 32] [--no-oracle]
"""
import argparse, ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from epos_amd import _lib, fitting
from helpers import fit_scenes as fs
ap = argparse.ArgumentParser()
ap.add_argument('--slots', type=int, default=32)
ap.add_argument('--no-oracle', action='store_true')
ap.add_argument('--gc-sweeps', type=int, default=None)
args = ap.parse_args()
lib = _lib.load()
K = fs.K_YCBV
def p(t): return ctypes.c_void_p(t.data_ptr())
def scene(rng, n, outlier):
  R = fs.rand_rot(rng); t = np.array([rng.uniform(-150, 150), rng.uniform(-100, 100), rng.uniform(600, 1200)])
  X = rng.uniform(-60, 60, (n, 3)); Y = X @ R.T + t
  q = Y @ K.T; q = q[:, :2] / q[:, 2:] + rng.standard_normal((n, 2))
  no = int(n * outlier); q[:no] = rng.uniform(0, [640, 480], (no, 2))
  o = np.argsort(q[:, 1], kind='stable')          # image-row order, as the pipeline hands it over
  return q[o], X[o], R, t
print('%6s %5s | %9s %12s %14s | %8s %8s | %10s' % ('N', 'outl', 'us/slot', 'hyp*pts/s', 'roots*pts/s', 'rot deg', 't mm', 'oracle ms'))
for n in (500, 2000, 10000):
  for outl in (0.3, 0.5, 0.7):
    rng = np.random.RandomState(n + int(outl * 10))
    S = args.slots
    sc = [scene(rng, n, outl) for _ in range(S)]
    xy = torch.from_numpy(np.concatenate([s[0] for s in sc])).cuda()
    xyz = torch.from_numpy(np.concatenate([s[1] for s in sc])).cuda()
    base = torch.arange(S + 1, dtype=torch.int64, device='cuda') * n
    Ks = torch.from_numpy(np.tile(K.reshape(9), (S, 1))).cuda()
    mm = torch.ones(S, dtype=torch.int32, device='cuda')
    seeds = torch.arange(S, dtype=torch.int64, device='cuda') + 7
    fp = fitting.fit_params(gc_sweeps=args.gc_sweeps)
    wb = lib.epos_fit_workspace_bytes(S, S * n, ctypes.byref(fp), 1)
    work = torch.empty(wb, dtype=torch.uint8, device='cuda')
    poses = torch.zeros(S, 12, dtype=torch.float64, device='cuda'); scores = torch.zeros(S, dtype=torch.float64, device='cuda')
    nm = torch.zeros(S, dtype=torch.int32, device='cuda'); labels = torch.zeros(S * n, dtype=torch.int32, device='cuda')
    def run():
      _lib.check(lib.epos_find6d_poses_device(p(xy), p(xyz), p(base), S, S * n, p(Ks), p(mm), p(seeds), ctypes.byref(fp), 1,
                                              p(work), p(poses), p(scores), p(nm), p(labels), None))
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    P = poses.cpu().numpy(); k = nm.cpu().numpy()
    errs = [fs.pose_err_sym(P[i, :9].reshape(3, 3), P[i, 9:], sc[i][2], sc[i][3]) for i in range(S) if k[i]]
    rot = np.median([e[0] for e in errs]) if errs else float('nan'); tr = np.median([e[1] for e in errs]) if errs else float('nan')
    oms = float('nan')
    if not args.no_oracle:
      from oracle import pnp_ref
      t0 = time.time()
      for i in range(min(S, 4)):
        pnp_ref.find6DPoses(sc[i][0], sc[i][1], K, seed=7 + i, params=pnp_ref.default_params(
            **({} if args.gc_sweeps is None else {'gc_sweeps': args.gc_sweeps})))
      oms = (time.time() - t0) / min(S, 4) * 1e3
    hp = S * 400 * n / (ms * 1e-3)
    print('%6d %5.0f%% | %9.1f %12.3e %14.3e | %8.3f %8.2f | %10.1f   found %d/%d' % (
        n, outl * 100, ms * 1e3 / S, hp, hp * 4, rot, tr, oms, int((k > 0).sum()), S))
