#!/usr/bin/env python
"""Per-workgroup timeline of pointwise_gemm_h2_f32 (a -DEPOS_GEMM_TRACE build): entry ->
prologue issued -> first stage landed (K loop starts) -> K loop done -> stores acknowledged,
and how much of a back-to-back launch (HIP events) lies outside the kernel's own span.

    python tools/gemm_h2_trace.py build      # here (cross-compile)
    python tools/gemm_h2_trace.py [--evict] [M N K [res]]   # on the GPU box
(--evict: a launch of another kernel variant before the traced one: cold instruction caches)
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epos_amd import build
PATH = os.path.join(build.LIB_DIR, 'libepos_gemm_h2_trace.so')
if len(sys.argv) > 1 and sys.argv[1] == 'build':
  subprocess.check_call([build.HIPCC] + build.FLAGS + ['-Wno-inline-asm', '-DEPOS_GEMM_TRACE'] +
                        os.environ.get('TRACE_DEFS', '').split() + ['-o', PATH] + build.sources())
  sys.exit(0)
import numpy as np, torch
from epos_amd import _lib
EVICT = '--evict' in sys.argv
sys.argv = [x for x in sys.argv if x != '--evict']
build.LIB_PATH = PATH
_lib._EVICT = '--evict' in sys.argv
sys.argv = [x for x in sys.argv if x != '--evict']
build.LIB_PATH = PATH
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
shapes = [(4800, 728, 728, 0), (4800, 728, 728, 1), (19200, 256, 256, 0), (4800, 2048, 1536, 0), (19200, 4032, 256, 0)]
if len(sys.argv) >= 4:
  shapes = [tuple(int(x) for x in sys.argv[1:4]) + (int(sys.argv[4]) if len(sys.argv) > 4 else 0,)]
raw = ctypes.CDLL(PATH)
for (m, n, k, res) in shapes:
  A = torch.relu(torch.randn(m, k, device='cuda')); C = torch.empty(m, n, device='cuda')
  R = torch.randn(m, n, device='cuda')
  slot = torch.zeros(64, dtype=torch.int32, device='cuda'); cs = torch.zeros(64, dtype=torch.int32, device='cuda')
  slot[0] = int(np.float32(8.0).view(np.int32))
  w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
  tot = lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, None); d8 = np.empty(tot, np.uint8)
  lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
  Wh = torch.from_numpy(d8).cuda()
  bias = torch.randn((n + 127) // 128 * 128, device='cuda')
  a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Wh), bias=p(bias), R=p(R) if res else None, ldr=n, C=p(C), ldc=n,
                         M=m, N=n, K=k, relu=1, relu_in=0, sub=1, Wh=p(Wh), a_amax=p(slot), c_amax=p(cs))
  nwg = ((m + 127) // 128) * ((n + 127) // 128)
  tr = torch.zeros(8 * nwg, dtype=torch.int64, device='cuda')
  raw.epos_debug_set_gemm_trace(ctypes.c_void_p(0))
  for _ in range(50): lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50): lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 50 * 1e3
  if EVICT:
    # the traced launch follows a launch of ANOTHER variant of the kernel over the whole chip
    # (residual epilogue, fp32 A: ~40 KB of different code through every instruction cache),
    # as in the plan, where a depthwise launch sits between two GEMMs
    R2 = torch.randn(m, n, device='cuda'); C2 = torch.empty(m, n, device='cuda')
    ev = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Wh), bias=p(bias), R=p(R2) if not res else None, ldr=n,
                            C=p(C2), ldc=n, M=m, N=n, K=k, relu=1, relu_in=0, sub=1, Wh=p(Wh),
                            a_amax=p(slot), c_amax=p(cs))
    for _ in range(3):
      raw.epos_debug_set_gemm_trace(ctypes.c_void_p(0))
      lib.epos_pointwise_conv_f32(ctypes.byref(ev), None)
      torch.cuda.synchronize()
      raw.epos_debug_set_gemm_trace(ctypes.c_void_p(tr.data_ptr()))
      lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
      torch.cuda.synchronize()
  else:
    raw.epos_debug_set_gemm_trace(ctypes.c_void_p(tr.data_ptr()))
    for _ in range(3): lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
    torch.cuda.synchronize()
  raw.epos_debug_set_gemm_trace(ctypes.c_void_p(0))
  t = tr.cpu().numpy().reshape(-1, 8)
  t = t[t[:, 0] > 0]
  t0 = t[:, 0].min()
  span = (t[:, 4].max() - t0) / 100.0
  print('%d x %d x %d%s: %d workgroups, back-to-back launch %.1f us (events), kernel span (first entry -> last store ack) %.1f us, %.1f TFLOP/s' % (
      m, n, k, ' + res' if res else '', len(t), us, span, 2.0 * m * n * k / us / 1e6))
  print('  entry skew: mean %.2f max %.2f us' % (((t[:, 0] - t0) / 100).mean(), ((t[:, 0] - t0) / 100).max()))
  for i, name in enumerate(['set-up + prologue DMA issue', 'first stage landed (scale, barrier)', 'K loop', 'epilogue (to store ack)']):
    d = (t[:, i + 1] - t[:, i]) / 100.0
    print('  %-38s mean %6.2f  max %6.2f us' % (name, d.mean(), d.max()))
  for a, b, name in [(3, 5, 'epilogue: operands + workgroup barrier'), (5, 6, 'epilogue: scale + bias -> LDS'),
                     (6, 7, 'epilogue: LDS rows -> (residual, ReLU) -> stores issued'), (7, 4, 'epilogue: absmax publish + store ack')]:
    d = (t[:, b] - t[:, a]) / 100.0
    print('    %-52s mean %6.2f  max %6.2f us' % (name, d.mean(), d.max()))
  print('  workgroup lifetime                     mean %6.2f  max %6.2f us' % (((t[:, 4] - t[:, 0]) / 100).mean(), ((t[:, 4] - t[:, 0]) / 100).max()))
