#!/usr/bin/env python
"""Steady-state rate of the split-operand GEMM's K loop: long K and whole rounds of tiles
(1024 tiles = 2 per workgroup slot), next to the same tile counts at the network's K."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
for (m, n, k) in [(16384, 1024, 4096), (16384, 1024, 728), (16384, 1024, 256), (8192, 1024, 4096), (19200, 728, 728)]:
  A = torch.randn(m, k, device='cuda'); C = torch.empty(m, n, device='cuda')
  w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
  tot = lib.epos_pack_pointwise_weights_split(None, k, n, None); d8 = np.empty(tot, np.uint8)
  lib.epos_pack_pointwise_weights_split(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
  Ws = torch.from_numpy(d8).cuda()
  a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Ws), bias=None, R=None, ldr=n, C=p(C), ldc=n, M=m, N=n, K=k, relu=0, relu_in=0, sub=1, Ws=p(Ws))
  call = lambda: lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
  for _ in range(30): call()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): call()
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 20 * 1e3
  print('%dx%dx%d tiles %d  %.1f us  %.1f TF' % (m, n, k, (m // 128) * (n // 128), us, 2 * m * n * k / us / 1e6))
