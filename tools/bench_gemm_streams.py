#!/usr/bin/env python
"""How does the fp16-pair GEMM scale with independent launches in flight? 1..6 streams
round-robin at the middle-flow shape (228 workgroups per launch, two resident per CU): the
aggregate algorithmic TFLOP/s the kernel reaches when the chip is kept full -- the ceiling
of what the pipelined step can get out of it."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
NS = 6
streams = [torch.cuda.Stream() for _ in range(NS)]
SHAPES = [(4800, 728, 728, 0), (4800, 728, 728, 1), (4800, 1024, 728, 0), (19200, 256, 304 + 16, 0)]
if len(sys.argv) > 1:          # e.g. 29184x128x728 (one column tile: no A sharing between workgroups)
  SHAPES = [tuple(int(v) for v in a.split('x')) + (0,) for a in sys.argv[1:]]
for (m, n, k, res) in SHAPES:
  As = [torch.relu(torch.randn(m, k, device='cuda')) for _ in range(NS)]
  Cs = [torch.empty(m, n, device='cuda') for _ in range(NS)]
  Rs = [torch.randn(m, n, device='cuda') for _ in range(NS)]
  slot = torch.zeros(64, dtype=torch.int32, device='cuda')
  slot[0] = int(np.float32(8.0).view(np.int32))
  w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
  tot = lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, None)
  d8 = np.empty(tot, np.uint8)
  lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
  Wh = torch.from_numpy(d8).cuda()
  args = [_lib.PointwiseArgs(A=p(As[i]), lda=k, Wp=p(Wh), bias=None, R=p(Rs[i]) if res else None, ldr=n,
                             C=p(Cs[i]), ldc=n, M=m, N=n, K=k, relu=0, relu_in=0, sub=1, Wh=p(Wh),
                             a_amax=p(slot)) for i in range(NS)]
  out = []
  for nstream in range(1, NS + 1):
    def call(i):
      st = streams[i % nstream]
      _lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(args[i % nstream]), ctypes.c_void_p(st.cuda_stream)))
    for i in range(240): call(i)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for st in streams: st.wait_stream(torch.cuda.current_stream())
    e0.record()
    for st in streams[:nstream]: st.wait_event(e0)
    for i in range(240): call(i)
    for st in streams[:nstream]: torch.cuda.current_stream().wait_stream(st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 240 * 1e3
    out.append('%d: %5.1f' % (nstream, 2 * m * n * k / us / 1e6))
  print('%dx%dx%d%s  TFLOP/s by streams in flight  %s' % (m, n, k, ' +res' if res else '', '  '.join(out)), flush=True)
