#!/usr/bin/env python
"""Per-workgroup timeline of the fused separable conv on the fp16-pair kernel
(epos_separable_conv_f32 with fp16-pair intermediates): a -DEPOS_SEPCONV_TRACE copy of the
library prints, for one shape, the average 100 MHz-stamp intervals of the producer phase
(depthwise compute, store drain + barrier, wait for the siblings, acquire) and the K loop,
next to the two-launch path's event times.

    python tools/sepconv_h2_trace.py build      # here (cross-compile)
    python tools/sepconv_h2_trace.py [b h w cin cout rate]     # on the GPU box
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from epos_amd import build
PATH = os.path.join(build.LIB_DIR, 'libepos_sepconv_h2_trace.so')
if len(sys.argv) > 1 and sys.argv[1] == 'build':
  srcs = build.sources()
  subprocess.check_call([build.HIPCC] + build.FLAGS + ['-Wno-inline-asm', '-DEPOS_SEPCONV_TRACE'] + os.environ.get('TRACE_DEFS', '').split() + ['-o', PATH] + srcs)
  sys.exit(0)
import numpy as np, torch
from epos_amd import _lib
build.LIB_PATH = PATH
_lib._build.LIB_PATH = PATH
lib = _lib.load()
import test_gpu_h2 as th
shape = [int(x) for x in sys.argv[1:7]] if len(sys.argv) >= 7 else [1, 60, 80, 728, 728, 2]
b, h, w, cin, cout, rate = shape
t, m, make = th._sepconv_h2_problem(lib, b, h, w, cin, cout, rate, 1, 0, 1)
sync = torch.zeros(int(lib.epos_separable_conv_sync_words(m)), dtype=torch.int32, device='cuda')
nwg = 8192
stats = torch.zeros(16 + 16 * nwg, dtype=torch.int32, device='cuda')
T = torch.zeros(m, cin, device='cuda'); C = torch.zeros(m, cout, device='cuda')
dw, pw, sa = make(T, C, sync, stats)
def timeit(fn, n=40):
  for _ in range(30): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
def two():
  _lib.check(lib.epos_depthwise3x3_f32(ctypes.byref(dw), None))
  _lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(pw), None))
t_dw = timeit(lambda: _lib.check(lib.epos_depthwise3x3_f32(ctypes.byref(dw), None)))
t_pw = timeit(lambda: _lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(pw), None)))
t_two = timeit(two)
t_f = timeit(lambda: _lib.check(lib.epos_separable_conv_f32(ctypes.byref(sa), None)))
tr = stats.cpu().numpy().view(np.int64)[8:].reshape(-1, 8)
tr = tr[tr[:, 0] > 0]
t0 = tr[:, 0].min()
names = ['depthwise produce', 'drain + barrier', 'wait for siblings', 'acquire + re-arm', 'K loop']
print('shape %s: %d workgroups; fused launch %.1f us; depthwise (fp16 pairs out) %.1f + pre-split GEMM %.1f, back to back %.1f us; time-outs %d' % (
    shape, len(tr), t_f, t_dw, t_pw, t_two, int(stats[0])))
print('start skew: mean %.2f us, max %.2f us' % ((tr[:, 0] - t0).mean() / 100, (tr[:, 0] - t0).max() / 100))
ok = tr[:, 6] > 0
print('producer: set-up %.2f us, first chunk landed after %.2f us, remaining chunks + stores %.2f us' % (
    ((tr[ok, 6] - tr[ok, 0]) / 100).mean(), ((tr[ok, 7] - tr[ok, 6]) / 100).mean(), ((tr[ok, 1] - tr[ok, 7]) / 100).mean()))
for i, n in enumerate(names):
  dlt = (tr[:, i + 1] - tr[:, i]) / 100.0
  print('%-20s mean %6.2f us  max %6.2f us' % (n, dlt.mean(), dlt.max()))
