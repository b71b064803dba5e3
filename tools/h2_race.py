#!/usr/bin/env python
"""Concurrency stress of the fp16-pair GEMM: the same problems solo and on four streams at
once, bit for bit. Modes:
  slot   -- explicit, precomputed absmax slots (isolates the kernel from the slot machinery)
  chain  -- GEMM1 (publishes c_amax) -> GEMM2 (reads it as a_amax), per stream, slot cleared
            before every step (the plan's protocol)
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
def pack(w, which):
  k, n = w.shape
  fn = {'p': lib.epos_pack_pointwise_weights, 'h2': lib.epos_pack_pointwise_weights_h2}[which]
  wp = w.ctypes.data_as(ctypes.c_void_p)
  tot = fn(wp, k, n, None)
  d = np.empty(tot, np.float32 if which == 'p' else np.uint8)
  fn(wp, k, n, d.ctypes.data_as(ctypes.c_void_p))
  return torch.from_numpy(d).cuda()
rng = np.random.RandomState(0)
NS = 4
streams = [torch.cuda.Stream() for _ in range(NS)]
bad = 0
for (m, k, n, n2) in [(4800, 728, 728, 728), (19200, 128, 256, 256), (3072, 256, 1344, 256)]:
  w1 = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  w2 = (rng.standard_normal((n, n2)) / np.sqrt(n)).astype(np.float32)
  W1p, W1h, W2p, W2h = pack(w1, 'p'), pack(w1, 'h2'), pack(w2, 'p'), pack(w2, 'h2')
  A = [torch.relu(torch.randn(m, k, device='cuda')) * (1 + i) for i in range(NS)]
  T = [torch.zeros(m, n, device='cuda') for _ in range(NS)]
  C = [torch.zeros(m, n2, device='cuda') for _ in range(NS)]
  sa = [torch.zeros(64, dtype=torch.int32, device='cuda') for _ in range(NS)]
  st = [torch.zeros(64, dtype=torch.int32, device='cuda') for _ in range(NS)]
  for i in range(NS):
    _lib.check(lib.epos_absmax_f32(p(A[i]), k, m, k, p(sa[i]), None))
  torch.cuda.synchronize()
  def step(i, stream):
    s = ctypes.c_void_p(stream.cuda_stream) if stream is not None else None
    _lib.check(lib.epos_amax_clear(p(st[i]), 1, s))
    a1 = _lib.PointwiseArgs(A=p(A[i]), lda=k, Wp=p(W1p), bias=None, R=None, ldr=0, C=p(T[i]), ldc=n,
                            M=m, N=n, K=k, relu=1, relu_in=0, sub=1, Wh=p(W1h), a_amax=p(sa[i]),
                            c_amax=p(st[i]))
    _lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(a1), s))
    a2 = _lib.PointwiseArgs(A=p(T[i]), lda=n, Wp=p(W2p), bias=None, R=None, ldr=0, C=p(C[i]), ldc=n2,
                            M=m, N=n2, K=n, relu=0, relu_in=0, sub=1, Wh=p(W2h), a_amax=p(st[i]))
    _lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(a2), s))
  ref = []
  for i in range(NS):
    step(i, None)
    torch.cuda.synchronize()
    ref.append((T[i].clone(), C[i].clone(), st[i].clone()))
  for rnd in range(60):
    for i in range(NS):
      step(i, streams[i])
    torch.cuda.synchronize()
    for i in range(NS):
      okT, okC, okS = torch.equal(T[i], ref[i][0]), torch.equal(C[i], ref[i][1]), torch.equal(st[i].max(), ref[i][2].max())
      if not (okT and okC and okS):
        bad += 1
        if bad < 10:
          dT = (T[i] != ref[i][0]).sum().item(); dC = (C[i] != ref[i][1]).sum().item()
          print('shape', (m, k, n, n2), 'round', rnd, 'stream', i, 'T diff', dT, 'C diff', dC, 'slot', okS,
                'maxabs dC', float((C[i] - ref[i][1]).abs().max()))
print('mismatching (round, stream) pairs:', bad)
