#!/usr/bin/env python
"""128 x 128 against 128 x 64 output tiles of the fp16-pair GEMM (the eight-wave form of round 4 and the
256 x 128 tile of round 5 were measured with this script at commits 0698228 / 1b05025 -- profiles/r04/gemm_h2_tile_shapes.txt,
profiles/r05/gemm_h2_tall_tile_shapes*.txt -- and left the library again). Round 4, judge's item 3:
"a 64-row tile (456 tiles)" for a single image's M = 4800 layers), per shape: one launch at
a time on one stream, and 2 / 4 streams round robin (what a pipeline of images looks like).
With and without pre-split A, residual epilogue for the middle-flow shape.

    python tools/bench_gemm_h2_tiles.py [128x128,128x64]     # on the GPU box
(EPOS_HIP_LIB=<variant> selects another build of the library)
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
PLAN = '--plan-args' in sys.argv     # bias, second absmax slot + gain, published output absmax (as the plan's layers)
PS = '--presplit' in sys.argv          # A already as fp16 pairs (what the plan's depthwise layers write)
argv = [a for a in sys.argv[1:] if not a.startswith('--')]
MODES = argv[0].split(',') if argv else ['128x128', '128x64']
def p(t): return ctypes.c_void_p(t.data_ptr())
# (M, N, K, residual)
BIG = '--big' in sys.argv            # launches of many tiles (batches of four / eight, the heads, steady state)
shapes = [(19200, 728, 728, 0), (19200, 728, 728, 1), (19200, 1024, 728, 0), (19200, 2048, 1536, 0), (19200, 256, 2048, 0),
          (76800, 4032, 256, 0), (19200, 4032, 256, 0), (76800, 256, 304, 0), (38400, 728, 728, 1), (16384, 1024, 4096, 0),
          (4800, 728, 728, 0), (4800, 2048, 1536, 0)] if BIG else [(4800, 728, 728, 0), (4800, 728, 728, 1), (4800, 1024, 728, 0), (4800, 1536, 1024, 0),
          (4800, 256, 2048, 0), (19200, 256, 304, 0), (19200, 256, 256, 0), (1200, 1024, 1536, 0)]
NS = 4
streams = [torch.cuda.Stream() for _ in range(NS)]
print('%-26s %-9s %s' % ('M x N x K', 'tile', 'us per launch at 1 / 2 / 4 streams   (TFLOP/s fp32-equivalent)'))
for (m, n, k, res) in shapes:
  As = [torch.relu(torch.randn(m, k, device='cuda')) for _ in range(NS)]
  if PS:      # any finite fp16 pairs do for timing: [4 hi | 4 mid] per 16 bytes
    As = [torch.randn(m, 2 * k, device='cuda').to(torch.float16).view(torch.float32) for _ in range(NS)]
  Cs = [torch.empty(m, n, device='cuda') for _ in range(NS)]
  R = torch.randn(m, n, device='cuda') if res else None
  slot = torch.zeros(64, dtype=torch.int32, device='cuda')
  slot[0] = int(np.float32(8.0).view(np.int32))
  w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
  tot = lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, None)
  if tot <= 0:      # a weight below the fp16-pair window of its column: the packer refuses
    print('%dx%dx%d: weights refused by the fp16-pair packer, skipped' % (m, n, k)); continue
  d8 = np.empty(tot, np.uint8)
  lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
  Wh = torch.from_numpy(d8).cuda()
  bias = torch.zeros((n + 127) // 128 * 128, device='cuda')
  slot2 = slot.clone(); cslots = [torch.zeros(64, dtype=torch.int32, device='cuda') for _ in range(NS)]
  extra = [dict(bias=p(bias), a_amax2=p(slot2), a_gain=1.5, a_bias=0.25, c_amax=p(cslots[i])) if PLAN else dict(bias=None) for i in range(NS)]
  args = [_lib.PointwiseArgs(A=p(As[i]), lda=k, Wp=p(Wh), **extra[i], R=p(R) if res else None, ldr=n,
                             C=p(Cs[i]), ldc=n, M=m, N=n, K=k, relu=1, relu_in=0, sub=1, Wh=p(Wh),
                             a_amax=p(slot), a_presplit=1 if PS else 0) for i in range(NS)]
  tiles = -(-m // 128) * -(-n // 128)
  for name, limit in (('128x128', 0), ('128x64', 1 << 30)):
    if name not in MODES: continue
    lib.epos_set_h2_narrow_tile_limit(limit)
    cells = []
    for nstream in (1, 2, 4):
      def call(i):
        st = streams[i % nstream]
        _lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(args[i % NS]), ctypes.c_void_p(st.cuda_stream)))
      reps = 120 if m * n * k < 2e11 else 40
      for i in range(reps): call(i)
      torch.cuda.synchronize()
      e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
      e0.record()
      for st in streams[:nstream]: st.wait_event(e0)
      for i in range(reps): call(i)
      for st in streams[:nstream]: torch.cuda.current_stream().wait_stream(st)
      e1.record(); torch.cuda.synchronize()
      us = e0.elapsed_time(e1) / reps * 1e3
      cells.append('%6.1f (%3.0f)' % (us, 2.0 * m * n * k / us * 1e-6))
    print('%-26s %-9s %s' % ('%dx%dx%d%s [%d]' % (m, n, k, '+R' if res else '', tiles), name, '   '.join(cells)), flush=True)
lib.epos_set_h2_narrow_tile_limit(100)
