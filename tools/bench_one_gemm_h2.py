#!/usr/bin/env python
"""ONE shape of the fp16-pair GEMM, a few launches (for rocprofv3 --pmc passes, tools/pmc_h2_gemm.sh):
    python tools/bench_one_gemm_h2.py M N K iters [presplit]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
m, n, k, it = [int(x) for x in sys.argv[1:5]]
ps = 'presplit' in sys.argv
lib.epos_set_h2_narrow_tile_limit(0)
A = torch.relu(torch.randn(m, k, device='cuda'))
if ps: A = torch.randn(m, 2 * k, device='cuda').to(torch.float16).view(torch.float32)
C = torch.empty(m, n, device='cuda')
w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
tot = lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, None)
d8 = np.empty(tot, np.uint8)
lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
Wh = torch.from_numpy(d8).cuda()
slot = torch.zeros(64, dtype=torch.int32, device='cuda'); slot[0] = int(np.float32(8.0).view(np.int32))
cs = torch.zeros(64, dtype=torch.int32, device='cuda')
b = torch.zeros((n + 127) // 128 * 128, device='cuda')
a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Wh), bias=p(b), R=None, ldr=n, C=p(C), ldc=n, M=m, N=n, K=k, relu=1,
                       relu_in=0, sub=1, Wh=p(Wh), a_amax=p(slot), a_presplit=1 if ps else 0, c_amax=p(cs))
for _ in range(it): _lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(a), None))
torch.cuda.synchronize()
if '--copy' in sys.argv:          # a plain copy of A beside the GEMM (PMC calibration)
  B2 = torch.empty_like(A)
  for _ in range(it): B2.copy_(A)
  torch.cuda.synchronize()
if '--time' in sys.argv:          # warm launches back to back, HIP events (round 6: EPOS_H2_PERSIST A/B)
  for _ in range(200): lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(200): lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 200 * 1e3
  print('%d x %d x %d%s: %.1f us per launch, %.0f TFLOP/s fp32-equivalent' % (
      m, n, k, ' presplit' if ps else '', us, 2.0 * m * n * k / us / 1e6))
