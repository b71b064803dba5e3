#!/usr/bin/env python
"""Ablations of the split-operand GEMM loop (what bounds it): builds copies of the
GEMM translation units with -DEPOS_SPLIT_ABL_{NODMA,NOBAR,NOREAD,NOSPLIT} next to the
real library (results of those are wrong by construction) and times a few shapes.

    python tools/bench_gemm_split_abl.py build      # here (cross-compile)
    python tools/bench_gemm_split_abl.py            # on the GPU box
"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epos_amd import build
VARIANTS = [[], ['-DEPOS_SPLIT_ABL_NODMA'], ['-DEPOS_SPLIT_ABL_NOBAR', '-DEPOS_SPLIT_ABL_NODMA'],
            ['-DEPOS_SPLIT_ABL_NOSPLIT'], ['-DEPOS_SPLIT_ABL_NOREAD'],
            ['-DEPOS_SPLIT_ABL_NODMA', '-DEPOS_SPLIT_ABL_NOBAR', '-DEPOS_SPLIT_ABL_NOREAD'],
            ['-DEPOS_SPLIT_ABL_NODMA', '-DEPOS_SPLIT_ABL_NOBAR', '-DEPOS_SPLIT_ABL_NOREAD', '-DEPOS_SPLIT_ABL_NOSPLIT']]
def path(defs):
  return os.path.join(build.LIB_DIR, 'libepos_abl%s.so' % ''.join(d.replace('-DEPOS_SPLIT_ABL', '').replace('-DEPOS_SPLIT', '') for d in defs))
if os.environ.get('ABL_VARIANTS'):      # e.g. ABL_VARIANTS=';-DEPOS_SPLIT_M0SAVE;;-DEPOS_SPLIT_M0SAVE'
  VARIANTS = [v.split() for v in os.environ['ABL_VARIANTS'].split(';')]
if len(sys.argv) > 1 and sys.argv[1] == 'build':
  srcs = [os.path.join(build.CSRC, f) for f in ('pointwise_gemm_split.hip', 'pointwise_gemm_dma.hip', 'pointwise_gemm.hip', 'layers.hip', 'runtime.hip')]
  for defs in [list(x) for x in {tuple(v) for v in VARIANTS}]:
    subprocess.check_call([build.HIPCC] + build.FLAGS + defs + ['-o', path(defs)] + srcs)
  sys.exit(0)
import numpy as np, torch
from epos_amd import _lib
def p(t): return ctypes.c_void_p(t.data_ptr())
shapes = [(4800, 728, 728), (19200, 728, 728), (19200, 4032, 256), (4800, 2048, 1536)]
for defs in VARIANTS:
  lib = ctypes.CDLL(path(defs))
  lib.epos_pack_pointwise_weights.restype = ctypes.c_int64
  lib.epos_pack_pointwise_weights_split.restype = ctypes.c_int64
  out = []
  for (m, n, k) in shapes:
    A = torch.randn(m, k, device='cuda'); C = torch.empty(m, n, device='cuda')
    w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
    tot = lib.epos_pack_pointwise_weights_split(None, k, n, None); d8 = np.empty(tot, np.uint8)
    lib.epos_pack_pointwise_weights_split(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
    Ws = torch.from_numpy(d8).cuda()
    a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Ws), bias=None, R=None, ldr=n, C=p(C), ldc=n,
                           M=m, N=n, K=k, relu=0, relu_in=0, sub=1, Ws=p(Ws))
    call = lambda: lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
    for _ in range(200): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    out.append('%dx%dx%d %6.1f us %5.1f TF' % (m, n, k, us, 2 * m * n * k / us / 1e6))
  print('%-40s %s' % (' '.join(d.replace('-DEPOS_SPLIT_ABL_', '') for d in defs) or 'full', ' | '.join(out)))
