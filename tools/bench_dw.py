#!/usr/bin/env python
"""Micro-benchmark of epos_depthwise3x3_f32 on the network's shapes (warm clocks),
next to a plain device copy of the same tensor (the streaming floor at that size).
EPOS_DW_L / EPOS_DW_THREADS select kernel variants."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from epos_amd import _lib
defs = os.environ.get('DW_DEFS', '').split()         # ablation builds of layers.hip
if defs:
  import subprocess
  from epos_amd import build
  path = os.path.join(build.LIB_DIR, 'libepos_hip_dw%s.so' % ''.join(d.replace('-D', '_') for d in defs))
  subprocess.check_call([build.HIPCC] + build.FLAGS + defs + ['-o', path] + build.sources())
  _lib.lib_path = lambda: path
lib = _lib.load()
H2 = '--h2' in sys.argv          # fp16-pair output (what the plan's depthwise layers write)
def p(t): return ctypes.c_void_p(t.data_ptr())
shapes = [(60, 80, 728, 2), (60, 80, 1024, 2), (60, 80, 1536, 4), (60, 80, 2048, 12), (120, 160, 256, 1),
          (120, 160, 304, 1), (240, 320, 64, 1), (240, 320, 128, 1), (60, 80, 256, 1)]
def timeit(fn, warm=200, it=100):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(it): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / it * 1e3
if '--one' in sys.argv:             # the middle-flow tensor only, few launches (PMC passes)
  shapes = shapes[:1]
  _t = timeit
  timeit = lambda fn, warm=3, it=5: _t(fn, warm, it)
PAD = '--pad32' in sys.argv       # rows padded to a multiple of 32 floats (128-byte lines)
for (h, w, c, rate) in shapes:
  ld = (c + 31) // 32 * 32 if PAD else c
  X = torch.randn(1, h, w, ld, device='cuda'); Y = torch.empty_like(X)
  w9 = torch.randn(9, c, device='cuda'); b = torch.randn(c, device='cuda')
  a = _lib.DepthwiseArgs(X=p(X), ldx=ld, w9c=p(w9), bias=p(b), Y=p(Y), ldy=ld, B=1, Hi=h, Wi=w,
                         Ho=h, Wo=w, C=c, stride=1, rate=rate, relu_in=1, relu_out=0)
  if H2:
    import numpy as np
    slot = torch.zeros(64, dtype=torch.int32, device='cuda'); slot[0] = int(np.float32(6.0).view(np.int32))
    a.x_amax = p(slot); a.gain = 30.0; a.bias0 = 3.0; a.y_h2 = 1
  us = timeit(lambda: lib.epos_depthwise3x3_f32(ctypes.byref(a), None))
  cp = timeit(lambda: Y.copy_(X))
  mb = 2 * X.numel() * 4 / 1e6
  print('%3dx%3dx%4d rate %2d  dw %6.1f us (%5.2f TB/s)   copy %6.1f us (%5.2f TB/s)' % (
      h, w, c, rate, us, mb / us / 1e6 * 1e6 / 1e6, cp, mb / cp))
