#!/usr/bin/env python
"""Upper bounds for a persistent tile loop with an overlapped epilogue (judge's item 1): the
fp16-pair kernel with its epilogue REMOVED (-DEPOS_H2_ABL_NOEPI: the most an overlapped
epilogue could hide) next to the real kernel, one launch at a time and two / three streams
round-robin, for the launches with many tiles per workgroup slot (heads, decoder) and the
middle-flow shape.

    python tools/bench_gemm_h2_bounds.py build      # here (cross-compile)
    python tools/bench_gemm_h2_bounds.py            # on the GPU box
"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epos_amd import build
VARIANTS = [('real kernel', []), ('no epilogue', ['-DEPOS_H2_ABL_NOEPI'])]
def path(defs):
  return os.path.join(build.LIB_DIR, 'libepos_h2bound%s.so' % ''.join(d.replace('-DEPOS_H2_ABL', '') for d in defs))
if len(sys.argv) > 1 and sys.argv[1] == 'build':
  srcs = [os.path.join(build.CSRC, f) for f in ('pointwise_gemm_h2.hip', 'pointwise_gemm_split.hip', 'pointwise_gemm_dma.hip', 'pointwise_gemm.hip', 'layers.hip', 'runtime.hip')]
  for _, defs in VARIANTS:
    subprocess.check_call([build.HIPCC] + build.FLAGS + ['-Wno-inline-asm'] + defs + ['-o', path(defs)] + srcs)
  sys.exit(0)
import numpy as np, torch
from epos_amd import _lib
def p(t): return ctypes.c_void_p(t.data_ptr())
shapes = [(19200, 4032, 256), (19200, 1344, 256), (19200, 256, 256), (4800, 728, 728), (4800, 2048, 1536)]
streams = [torch.cuda.Stream() for _ in range(3)]
for vname, defs in VARIANTS:
  lib = ctypes.CDLL(path(defs))
  lib.epos_pack_pointwise_weights_h2.restype = ctypes.c_int64
  out = []
  for (m, n, k) in shapes:
    As = [torch.relu(torch.randn(m, k, device='cuda')) for _ in range(3)]
    Cs = [torch.empty(m, n, device='cuda') for _ in range(3)]
    slot = torch.zeros(64, dtype=torch.int32, device='cuda')
    slot[0] = int(np.float32(8.0).view(np.int32))
    w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
    tot = lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, None); d8 = np.empty(tot, np.uint8)
    lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
    Wh = torch.from_numpy(d8).cuda()
    args = [_lib.PointwiseArgs(A=p(As[i]), lda=k, Wp=p(Wh), bias=None, R=None, ldr=n, C=p(Cs[i]), ldc=n,
                               M=m, N=n, K=k, relu=0, relu_in=0, sub=1, Wh=p(Wh), a_amax=p(slot)) for i in range(3)]
    res = []
    for nstream in (1, 2, 3):
      def call(i):
        st = streams[i % nstream]
        lib.epos_pointwise_conv_f32(ctypes.byref(args[i % 3]), ctypes.c_void_p(st.cuda_stream))
      for i in range(60): call(i)
      torch.cuda.synchronize()
      e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
      for st in streams: st.wait_stream(torch.cuda.current_stream())
      e0.record()
      for st in streams[:nstream]: st.wait_event(e0)
      for i in range(60): call(i)
      for st in streams[:nstream]: torch.cuda.current_stream().wait_stream(st)
      e1.record(); torch.cuda.synchronize()
      res.append('%6.1f' % (e0.elapsed_time(e1) / 60 * 1e3))
    out.append('%dx%dx%d %s us' % (m, n, k, ' / '.join(res)))
  print('%-12s %s' % (vname, ' | '.join(out)), flush=True)
