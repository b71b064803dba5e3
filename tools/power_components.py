#!/usr/bin/env python
"""Socket power and shader clock (rocm-smi) under each component of the step run ALONE in a
loop for a few seconds: idle, a plain device copy, the depthwise kernel, the split GEMM at the
network's dominant shape and at a steady-state shape. With the pipeline's own figure
(tools/power_sample.sh) this is the energy budget of an image: time x power per component."""
import ctypes, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())

samples, stop = [], [False]
def sampler():
  while not stop[0]:
    try:
      o = subprocess.run(['/opt/rocm/bin/rocm-smi', '--showpower', '--showclocks'],
                         capture_output=True, text=True, timeout=5).stdout
      w = re.search(r'Power \(W\): ([0-9.]+)', o)
      c = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', o)
      if w and c:
        samples.append((time.time(), float(w.group(1)), int(c.group(1))))
    except Exception:
      pass
    time.sleep(0.2)
th = threading.Thread(target=sampler, daemon=True); th.start()

def gemm(m, n, k):
  A = torch.randn(m, k, device='cuda'); C = torch.empty(m, n, device='cuda')
  w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
  tot = lib.epos_pack_pointwise_weights_split(None, k, n, None); d8 = np.empty(tot, np.uint8)
  lib.epos_pack_pointwise_weights_split(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
  Ws = torch.from_numpy(d8).cuda()
  a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Ws), bias=None, R=None, ldr=n, C=p(C), ldc=n, M=m, N=n, K=k, relu=0, relu_in=0, sub=1, Ws=p(Ws))
  keep = (A, C, Ws, a)
  return (lambda: lib.epos_pointwise_conv_f32(ctypes.byref(a), None)), keep, 2.0 * m * n * k
def dw(h, w_, c, rate):
  X = torch.randn(1, h, w_, c, device='cuda'); Y = torch.empty_like(X)
  w9 = torch.randn(9, c, device='cuda'); b = torch.randn(c, device='cuda')
  a = _lib.DepthwiseArgs(X=p(X), ldx=c, w9c=p(w9), bias=p(b), Y=p(Y), ldy=c, B=1, Hi=h, Wi=w_, Ho=h, Wo=w_, C=c, stride=1, rate=rate, relu_in=1, relu_out=0)
  keep = (X, Y, w9, b, a)
  return (lambda: lib.epos_depthwise3x3_f32(ctypes.byref(a), None)), keep, 2.0 * h * w_ * c * 4
def copy(nbytes):
  X = torch.empty(nbytes // 4, device='cuda'); Y = torch.empty_like(X)
  return (lambda: Y.copy_(X)), (X, Y), 2.0 * nbytes

def phase(name, call, unit_work, unit, secs=4.0):
  if call is None:
    time.sleep(secs); t0, t1, n = time.time() - secs, time.time(), 0
  else:
    for _ in range(50): call()
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
      for _ in range(200): call()
      torch.cuda.synchronize(); n += 200
    t1 = time.time()
  s = [(w, c) for (t, w, c) in samples if t0 + 1.0 < t < t1]
  pw = np.mean([x[0] for x in s]) if s else float('nan')
  ck = np.mean([x[1] for x in s]) if s else float('nan')
  rate = n * unit_work / (t1 - t0) if n else 0.0
  us = (t1 - t0) / n * 1e6 if n else 0.0
  print('%-46s %6.0f W  sclk %4.0f MHz  %7.1f us/launch  %8.1f %s  (%d samples)' % (
      name, pw, ck, us, rate / (1e12 if unit == 'TFLOP/s' else 1e12), unit, len(s)))

phase('idle', None, 0, '-')
c, k1, wk = copy(14 * 2**20 * 4); phase('device copy, 56 MB tensors', c, wk, 'TB/s')
c, k2, wk = copy(14 * 2**20); phase('device copy, 14 MB tensors (60x80x728)', c, wk, 'TB/s')
c, k3, wk = dw(60, 80, 728, 2); phase('depthwise 60x80x728 rate 2', c, wk, 'TB/s')
c, k4, wk = dw(120, 160, 256, 1); phase('depthwise 120x160x256', c, wk, 'TB/s')
c, k5, wk = gemm(4800, 728, 728); phase('split GEMM 4800x728x728 (middle flow)', c, wk, 'TFLOP/s')
c, k6, wk = gemm(19200, 728, 728); phase('split GEMM 19200x728x728', c, wk, 'TFLOP/s')
c, k7, wk = gemm(16384, 1024, 4096); phase('split GEMM 16384x1024x4096 (steady state)', c, wk, 'TFLOP/s')
phase('idle again', None, 0, '-')
stop[0] = True
