#!/usr/bin/env python
"""Do independent GEMM launches on different HIP streams fill each other's
tile-quantisation gaps? N launches of one shape on 1, 2, 3 streams (round-robin)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
m, n, k = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (4800, 728, 728)
nstream_max = 3
bufs = []
for s in range(nstream_max):
  A = torch.randn(m, k, device='cuda'); C = torch.empty(m, n, device='cuda')
  w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
  total = lib.epos_pack_pointwise_weights(None, k, n, None); dst = np.empty(total, np.float32)
  lib.epos_pack_pointwise_weights(w.ctypes.data_as(ctypes.c_void_p), k, n, dst.ctypes.data_as(ctypes.c_void_p))
  Wp = torch.from_numpy(dst).cuda(); b = torch.zeros((n + 127) // 128 * 128, device='cuda')
  a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Wp), bias=p(b), R=None, ldr=n, C=p(C), ldc=n, M=m, N=n, K=k,
                         relu=0, relu_in=0, sub=1)
  bufs.append((A, C, Wp, b, a))
streams = [torch.cuda.Stream() for _ in range(nstream_max)]
clk = torch.zeros((8, 2), dtype=torch.int64, device='cuda'); cs = torch.cuda.Stream()
N = 600
for ns in (1, 2, 3):
  for rep in range(2):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in streams[:ns]: s.wait_stream(torch.cuda.current_stream())
    for i in range(N):
      j = i % ns
      lib.epos_pointwise_conv_f32(ctypes.byref(bufs[j][4]), ctypes.c_void_p(streams[j].cuda_stream))
      if rep == 1 and i in (N // 2, N // 2 + 100):      # sample the core clock mid-run
        lib.epos_clock_probe(ctypes.c_void_p(clk[i // 100 % 8].data_ptr()), 300, ctypes.c_void_p(cs.cuda_stream))
    for s in streams[:ns]: torch.cuda.current_stream().wait_stream(s)
    e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / N * 1e3
  ch = clk.cpu().numpy(); ch = ch[ch[:, 1] > 0]; mhz = float((ch[:, 0] / ch[:, 1]).mean() * 100) if len(ch) else 0; clk.zero_()
  print('M=%d N=%d K=%d  %d stream(s): %.1f us per launch  %.1f TFLOP/s aggregate  core clock %.0f MHz (roof at that clock %.1f)' % (m, n, k, ns, us, 2 * m * n * k / us / 1e6, mhz, 65.536 * mhz / 1e3))
