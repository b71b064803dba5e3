#!/usr/bin/env python
"""Interference between independent GEMM launches (2 streams) and depthwise launches
(1 stream) running concurrently: aggregate GEMM TFLOP/s and depthwise rate, together
and alone."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
defs = os.environ.get('DW_DEFS', '').split()
if defs:
  import subprocess
  from epos_amd import build
  path = os.path.join(build.LIB_DIR, 'libepos_hip_dw%s.so' % ''.join(d.replace('-D', '_').replace('=', '') for d in defs))
  subprocess.check_call([build.HIPCC] + build.FLAGS + defs + ['-o', path] + build.sources())
  _lib.lib_path = lambda: path
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
m, n, k = 4800, 728, 728
def mk_gemm():
  A = torch.randn(m, k, device='cuda'); C = torch.empty(m, n, device='cuda')
  w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
  if os.environ.get('EPOS_GEMM_SPLIT', '1') != '0':          # the split-operand kernel
    total = lib.epos_pack_pointwise_weights_split(None, k, n, None); d8 = np.empty(total, np.uint8)
    lib.epos_pack_pointwise_weights_split(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
    Ws = torch.from_numpy(d8).cuda()
    a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Ws), bias=None, R=None, ldr=n, C=p(C), ldc=n, M=m, N=n, K=k,
                           relu=0, relu_in=0, sub=1, Ws=p(Ws))
    Wh = None
    if os.environ.get('EPOS_GEMM_H2', '1') != '0':           # the fp16-pair kernel (default)
      total = lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, None); d8 = np.empty(max(total, 1), np.uint8)
      if total > 0 and lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n,
                                            d8.ctypes.data_as(ctypes.c_void_p)) > 0:
        Wh = torch.from_numpy(d8).cuda()
        a.Wh = p(Wh)
    return (A, C, Ws, Wh, a)
  total = lib.epos_pack_pointwise_weights(None, k, n, None); dst = np.empty(total, np.float32)
  lib.epos_pack_pointwise_weights(w.ctypes.data_as(ctypes.c_void_p), k, n, dst.ctypes.data_as(ctypes.c_void_p))
  Wp = torch.from_numpy(dst).cuda(); b = torch.zeros((n + 127) // 128 * 128, device='cuda')
  a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Wp), bias=p(b), R=None, ldr=n, C=p(C), ldc=n, M=m, N=n, K=k,
                         relu=0, relu_in=0, sub=1)
  return (A, C, Wp, b, a)
def mk_dw():
  h, w_, c, rate = 60, 80, 728, 2
  X = torch.randn(1, h, w_, c, device='cuda'); Y = torch.empty_like(X)
  w9 = torch.randn(9, c, device='cuda'); b = torch.randn(c, device='cuda')
  a = _lib.DepthwiseArgs(X=p(X), ldx=c, w9c=p(w9), bias=p(b), Y=p(Y), ldy=c, B=1, Hi=h, Wi=w_,
                         Ho=h, Wo=w_, C=c, stride=1, rate=rate, relu_in=1, relu_out=0)
  return (X, Y, w9, b, a)
g = [mk_gemm(), mk_gemm()]; d = mk_dw()
sg = [torch.cuda.Stream(), torch.cuda.Stream()]; sd = torch.cuda.Stream()
def run(n_gemm, n_dw):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  ng = nd = 0
  # interleave submissions so that all queues stay fed
  steps = max(n_gemm, n_dw)
  for i in range(steps):
    if i < n_gemm:
      j = i % 2
      lib.epos_pointwise_conv_f32(ctypes.byref(g[j][4]), ctypes.c_void_p(sg[j].cuda_stream)); ng += 1
    if n_dw and i * n_dw // steps != (i + 1) * n_dw // steps:
      lib.epos_depthwise3x3_f32(ctypes.byref(d[4]), ctypes.c_void_p(sd.cuda_stream)); nd += 1
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) * 1e6, ng, nd
for label, a, b_ in [('gemm only (2 streams)', 2000, 0), ('dw only', 0, 2000), ('gemm + dw 1:1', 2000, 2000), ('gemm + dw 1:1', 2000, 2000)]:
  run(a // 4, b_ // 4)
  us, ng, nd = run(a, b_)
  msg = '%-24s %.0f us' % (label, us)
  if ng: msg += '  GEMM %.1f us/launch = %.1f TFLOP/s' % (us / ng, 2 * m * n * k * ng / us / 1e6)
  if nd: msg += '  dw %.1f us/launch' % (us / nd)
  print(msg)
