#!/usr/bin/env python
"""Socket power and sclk (rocm-smi) while ONE shape of the fp16-pair GEMM runs back to back for a
few seconds, next to the delivered TFLOP/s: python tools/steady_power.py [M N K] [presplit]"""
import ctypes, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
shapes = [tuple(nums[:3])] if len(nums) >= 3 else [(16384, 1024, 4096), (19200, 728, 728), (4800, 728, 728)]
ps = 'presplit' in sys.argv
def smi():
  o = subprocess.run(['/opt/rocm/bin/rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
  w = re.search(r'Power \(W\): ([0-9.]+)', o); c = re.search(r'sclk clock level[^(]*\(([0-9.]+)Mhz\)', o)
  return (float(w.group(1)) if w else None, float(c.group(1)) if c else None)
print('idle: %s W, sclk %s MHz' % smi())
for (m, n, k) in shapes:
  A = torch.relu(torch.randn(m, k, device='cuda'))
  if ps: A = torch.randn(m, 2 * k, device='cuda').to(torch.float16).view(torch.float32)
  C = torch.empty(m, n, device='cuda')
  w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
  tot = lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, None); d8 = np.empty(tot, np.uint8)
  lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
  Wh = torch.from_numpy(d8).cuda()
  slot = torch.zeros(64, dtype=torch.int32, device='cuda'); slot[0] = int(np.float32(8.0).view(np.int32))
  cs = torch.zeros(64, dtype=torch.int32, device='cuda'); b = torch.zeros((n + 127) // 128 * 128, device='cuda')
  a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Wh), bias=p(b), R=None, ldr=n, C=p(C), ldc=n, M=m, N=n, K=k, relu=1,
                         relu_in=0, sub=1, Wh=p(Wh), a_amax=p(slot), a_presplit=1 if ps else 0, c_amax=p(cs))
  streams = [torch.cuda.Stream(), torch.cuda.Stream()]
  samples, stop = [], [False]
  def sampler():
    while not stop[0]:
      samples.append(smi())
  th = threading.Thread(target=sampler, daemon=True); th.start()
  t0 = time.perf_counter(); n_launch = 0
  while time.perf_counter() - t0 < 4.0:
    for _ in range(200):
      st = streams[n_launch & 1]
      lib.epos_pointwise_conv_f32(ctypes.byref(a), ctypes.c_void_p(st.cuda_stream)); n_launch += 1
    torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  stop[0] = True; th.join(timeout=5)
  ws = [s[0] for s in samples[1:] if s[0]]; cl = [s[1] for s in samples[1:] if s[1]]
  print('%d x %d x %d%s, two streams: %.1f us per launch = %.0f TFLOP/s fp32-equivalent (%.0f fp16) | %.0f W (max %.0f), sclk %.0f MHz [%d samples]' % (
      m, n, k, ' presplit' if ps else '', dt / n_launch * 1e6, 2.0 * m * n * k * n_launch / dt / 1e12, 6.0 * m * n * k * n_launch / dt / 1e12,
      np.mean(ws), np.max(ws), np.mean(cl), len(ws)))
