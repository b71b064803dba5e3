#!/usr/bin/env python
"""Ablations of the fp16-pair GEMM loop (what bounds it): copies of the GEMM translation
units with -DEPOS_H2_ABL_{NODMA,NOBAR,NOREAD,NOSPLIT} next to the real library (results of
those are wrong by construction; the stale LDS contents are real data, not zeros) and a few
shapes, one launch at a time and two streams round-robin (two workgroups per CU, as in the
pipelined step).

    python tools/bench_gemm_h2_abl.py build      # here (cross-compile)
    python tools/bench_gemm_h2_abl.py            # on the GPU box
"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epos_amd import build
VARIANTS = [[], ['-DEPOS_H2_ABL_NOSPLIT'], ['-DEPOS_H2_ABL_NODMA'], ['-DEPOS_H2_ABL_NOREAD'],
            ['-DEPOS_H2_ABL_NOBAR', '-DEPOS_H2_ABL_NODMA'],
            ['-DEPOS_H2_ABL_NODMA', '-DEPOS_H2_ABL_NOBAR', '-DEPOS_H2_ABL_NOREAD', '-DEPOS_H2_ABL_NOSPLIT']]
def path(defs):
  return os.path.join(build.LIB_DIR, 'libepos_h2abl%s.so' % ''.join(d.replace('-DEPOS_H2_ABL', '') for d in defs))
if len(sys.argv) > 1 and sys.argv[1] == 'build':
  srcs = [os.path.join(build.CSRC, f) for f in ('pointwise_gemm_h2.hip', 'pointwise_gemm_split.hip', 'pointwise_gemm_dma.hip', 'pointwise_gemm.hip', 'layers.hip', 'runtime.hip')]
  for defs in VARIANTS:
    subprocess.check_call([build.HIPCC] + build.FLAGS + ['-Wno-inline-asm'] + defs + ['-o', path(defs)] + srcs)
  sys.exit(0)
import numpy as np, torch
from epos_amd import _lib
def p(t): return ctypes.c_void_p(t.data_ptr())
shapes = [(4800, 728, 728), (19200, 728, 728), (19200, 4032, 256), (4800, 2048, 1536)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for defs in VARIANTS:
  lib = ctypes.CDLL(path(defs))
  lib.epos_pack_pointwise_weights.restype = ctypes.c_int64
  lib.epos_pack_pointwise_weights_h2.restype = ctypes.c_int64
  out = []
  for (m, n, k) in shapes:
    As = [torch.relu(torch.randn(m, k, device='cuda')) for _ in range(2)]
    Cs = [torch.empty(m, n, device='cuda') for _ in range(2)]
    slot = torch.zeros(64, dtype=torch.int32, device='cuda')
    slot[0] = int(np.float32(8.0).view(np.int32))
    w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
    tot = lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, None); d8 = np.empty(tot, np.uint8)
    lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
    Wh = torch.from_numpy(d8).cuda()
    args = [_lib.PointwiseArgs(A=p(As[i]), lda=k, Wp=p(Wh), bias=None, R=None, ldr=n, C=p(Cs[i]), ldc=n,
                               M=m, N=n, K=k, relu=0, relu_in=0, sub=1, Wh=p(Wh), a_amax=p(slot)) for i in range(2)]
    res = []
    for nstream in (1, 2):
      def call(i):
        st = streams[i % nstream]
        lib.epos_pointwise_conv_f32(ctypes.byref(args[i % 2]), ctypes.c_void_p(st.cuda_stream))
      for i in range(200): call(i)
      torch.cuda.synchronize()
      e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
      for st in streams: st.wait_stream(torch.cuda.current_stream())
      e0.record()
      for st in streams[:nstream]: st.wait_event(e0)
      for i in range(100): call(i)
      for st in streams[:nstream]: torch.cuda.current_stream().wait_stream(st)
      e1.record(); torch.cuda.synchronize()
      us = e0.elapsed_time(e1) / 100 * 1e3
      res.append('%5.1f TF' % (2 * m * n * k / us / 1e6))
    out.append('%dx%dx%d %s' % (m, n, k, ' / '.join(res)))
  print('%-28s %s' % (' '.join(d.replace('-DEPOS_H2_ABL_', '') for d in defs) or 'full', ' | '.join(out)), flush=True)
