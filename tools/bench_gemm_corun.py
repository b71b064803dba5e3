#!/usr/bin/env python
"""What costs a GEMM its throughput when another kernel shares the CUs? Two streams
of back-to-back GEMMs next to a third stream of co-runners that only OCCUPY wave slots
(sleep), issue VALU, or stream memory -- 1 or 2 extra waves per SIMD."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
sl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'interf', 'libsleepers.so'))
def p(t): return ctypes.c_void_p(t.data_ptr())
m, n, k = 4800, 728, 728
def mk_gemm():
  A = torch.randn(m, k, device='cuda'); C = torch.empty(m, n, device='cuda')
  w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
  total = lib.epos_pack_pointwise_weights(None, k, n, None); dst = np.empty(total, np.float32)
  lib.epos_pack_pointwise_weights(w.ctypes.data_as(ctypes.c_void_p), k, n, dst.ctypes.data_as(ctypes.c_void_p))
  Wp = torch.from_numpy(dst).cuda(); b = torch.zeros((n + 127) // 128 * 128, device='cuda')
  a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Wp), bias=p(b), R=None, ldr=n, C=p(C), ldc=n, M=m, N=n, K=k,
                         relu=0, relu_in=0, sub=1)
  return (A, C, Wp, b, a)
g = [mk_gemm(), mk_gemm()]
sg = [torch.cuda.Stream(), torch.cuda.Stream()]; sc = torch.cuda.Stream()
buf = torch.randn(64 << 20, device='cuda'); out = torch.zeros(16, device='cuda')
def corun(kind, blocks, us):
  s = ctypes.c_void_p(sc.cuda_stream)
  if kind == 'sleep': sl.launch_sleep(blocks, us, s)
  elif kind == 'valu': sl.launch_valu(blocks, us, p(out), s)
  elif kind == 'mem': sl.launch_mem(blocks, us, p(buf), p(out), buf.numel() // 4, s)
def run(ngemm, kind, blocks, us=100):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(ngemm):
    j = i % 2
    lib.epos_pointwise_conv_f32(ctypes.byref(g[j][4]), ctypes.c_void_p(sg[j].cuda_stream))
    if kind and i % 2 == 0: corun(kind, blocks, us)     # ~one 100 us co-runner per two ~40 us GEMMs
  for s_ in sg: s_.synchronize()
  dt = (time.perf_counter() - t0) * 1e6
  torch.cuda.synchronize()
  return dt
for kind, blocks in [(None, 0), ('sleep', 256), ('sleep', 512), ('valu', 256), ('valu', 512), ('mem', 256), ('mem', 512)]:
  run(500, kind, blocks)
  us = run(2000, kind, blocks)
  print('%-6s blocks %4d (%d extra wave(s)/SIMD): GEMM %.1f us/launch = %.1f TFLOP/s' % (
      kind or 'none', blocks, blocks // 256, us / 2000, 2 * m * n * k * 2000 / us / 1e6))
