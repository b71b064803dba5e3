#!/usr/bin/env python
"""Per-workgroup timeline of the fused separable conv (epos_separable_conv_f32):
builds a -DEPOS_SEPCONV_TRACE copy of the GEMM translation units and prints, for one
shape, the average 100 MHz-stamp intervals of the producer phase: depthwise compute,
store drain + barrier, wait for the siblings, acquire, K loop.

    python tools/sepconv_trace.py build      # here (cross-compile)
    python tools/sepconv_trace.py [b h w cin cout rate]     # on the GPU box
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from epos_amd import build
PATH = os.path.join(build.LIB_DIR, 'libepos_sepconv_trace.so')
if len(sys.argv) > 1 and sys.argv[1] == 'build':
  srcs = build.sources()
  subprocess.check_call([build.HIPCC] + build.FLAGS + ['-DEPOS_SEPCONV_TRACE'] + os.environ.get('TRACE_DEFS', '').split() + ['-o', PATH] + srcs)
  sys.exit(0)
import numpy as np, torch
from epos_amd import _lib
build.LIB_PATH = PATH
_lib._build.LIB_PATH = PATH
lib = _lib.load()
import test_gpu_layers as tl
shape = [int(x) for x in sys.argv[1:7]] if len(sys.argv) >= 7 else [1, 60, 80, 728, 728, 2]
b, h, w, cin, cout, rate = shape
t, m, make = tl._sepconv_problem(lib, b, h, w, cin, cout, rate, 1, 0, 1)
sync = torch.zeros(int(lib.epos_separable_conv_sync_words(m)), dtype=torch.int32, device='cuda')
nwg = 4096
stats = torch.zeros(16 + 16 * nwg, dtype=torch.int32, device='cuda')
T = torch.zeros(m, cin, device='cuda'); C = torch.zeros(m, cout, device='cuda')
sa = make(T, C, sync)[2]; sa.stats = ctypes.c_void_p(stats.data_ptr())
for _ in range(30):
  _lib.check(lib.epos_separable_conv_f32(ctypes.byref(sa), None))
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
  _lib.check(lib.epos_separable_conv_f32(ctypes.byref(sa), None))
e1.record(); torch.cuda.synchronize()
tr = stats.cpu().numpy().view(np.int64)[8:].reshape(-1, 8)
tr = tr[tr[:, 0] > 0]
t0 = tr[:, 0].min()
names = ['depthwise compute', 'drain + barrier', 'wait for siblings', 'acquire + re-arm', 'K loop']
print('shape %s: %d workgroups, launch %.1f us (events), time-outs %d' % (shape, len(tr), e0.elapsed_time(e1) / 20 * 1e3, int(stats[0])))
print('start skew: mean %.2f us, max %.2f us' % ((tr[:, 0] - t0).mean() / 100, (tr[:, 0] - t0).max() / 100))
for i, n in enumerate(names):
  dlt = (tr[:, i + 1] - tr[:, i]) / 100.0
  print('%-20s mean %6.2f us  max %6.2f us' % (n, dlt.mean(), dlt.max()))
