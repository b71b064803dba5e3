#!/usr/bin/env python
"""Where the watts go in the fp16-pair GEMM (pointwise_gemm_h2_f32): socket power, shader
clock (rocm-smi) and delivered TFLOP/s of ablated builds of the kernel, each alone in a loop
for a few seconds -- MFMAs only -> + LDS-DMA -> + barrier / waits -> + fragment reads ->
+ operand split (= the full loop) -> + epilogue (= the real kernel) -- at the middle-flow
shape on two streams (two workgroups per CU, as in the pipelined step) and at a steady-state
shape. Ablated builds compute wrong results by construction (stale but REAL LDS contents,
not zeros: all-zero operands draw less power).

    python tools/power_components_h2.py build      # here (cross-compile)
    python tools/power_components_h2.py            # on the GPU box
"""
import ctypes, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epos_amd import build
A = '-DEPOS_H2_ABL_'
VARIANTS = [
    ('MFMAs only (no DMA, barrier, reads, split, epilogue)', ['NODMA', 'NOBAR', 'NOREAD', 'NOSPLIT', 'NOEPI']),
    ('+ LDS-DMA', ['NOBAR', 'NOREAD', 'NOSPLIT', 'NOEPI']),
    ('+ barrier and counted waits', ['NOREAD', 'NOSPLIT', 'NOEPI']),
    ('+ fragment reads (ds_read_b128)', ['NOSPLIT', 'NOEPI']),
    ('+ operand split = full loop, no epilogue', ['NOEPI']),
    ('+ epilogue = the real kernel', []),
]
def path(defs):
  return os.path.join(build.LIB_DIR, 'libepos_h2pw_%s.so' % ('_'.join(defs) or 'full'))
if len(sys.argv) > 1 and sys.argv[1] == 'build':
  srcs = [os.path.join(build.CSRC, f) for f in ('pointwise_gemm_h2.hip', 'pointwise_gemm_split.hip', 'pointwise_gemm_dma.hip', 'pointwise_gemm.hip', 'layers.hip', 'runtime.hip')]
  from concurrent.futures import ThreadPoolExecutor
  def one(v):
    subprocess.check_call([build.HIPCC] + build.FLAGS + ['-Wno-inline-asm'] + [A + d for d in v[1]] + ['-o', path(v[1])] + srcs)
  with ThreadPoolExecutor(max_workers=6) as ex:
    list(ex.map(one, VARIANTS))
  sys.exit(0)
import numpy as np, torch
from epos_amd import _lib
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
    time.sleep(0.15)
threading.Thread(target=sampler, daemon=True).start()

streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def phase(name, call, flops, secs=3.5):
  if call is None:
    time.sleep(secs); t0, t1, n = time.time() - secs, time.time(), 0
  else:
    for i in range(100): call(i)
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
      for i in range(400): call(i)
      torch.cuda.synchronize(); n += 400
    t1 = time.time()
  s = [(w, c) for (t, w, c) in samples if t0 + 1.0 < t < t1]
  pw = np.mean([x[0] for x in s]) if s else float('nan')
  ck = np.mean([x[1] for x in s]) if s else float('nan')
  tf = n * flops / (t1 - t0) / 1e12 if n else 0.0
  us = (t1 - t0) / n * 1e6 if n else 0.0
  print('  %-58s %6.0f W  sclk %4.0f MHz  %7.1f us/launch  %6.1f TFLOP/s fp32-eq (%6.0f fp16)  [%d samples]' % (
      name, pw, ck, us, tf, 3 * tf, len(s)), flush=True)
  return pw, ck, tf

phase('idle', None, 0)
shapes = [((4800, 728, 728), 2, 'middle flow, two streams'),
          ((19200, 728, 728), 1, '19200 rows, one stream'),
          ((16384, 1024, 4096), 1, 'steady state (K = 4096, 1024 tiles)')]
for (m, n, k), nstream, label in shapes:
  print('%d x %d x %d  (%s)' % (m, n, k, label), flush=True)
  for vname, defs in VARIANTS:
    lib = ctypes.CDLL(path(defs))
    lib.epos_pack_pointwise_weights_h2.restype = ctypes.c_int64
    As = [torch.relu(torch.randn(m, k, device='cuda')) for _ in range(2)]
    Cs = [torch.empty(m, n, device='cuda') for _ in range(2)]
    slot = torch.zeros(64, dtype=torch.int32, device='cuda')
    slot[0] = int(np.float32(8.0).view(np.int32))
    w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
    tot = lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, None)
    d8 = np.empty(tot, np.uint8)
    lib.epos_pack_pointwise_weights_h2(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
    Wh = torch.from_numpy(d8).cuda()
    args = [_lib.PointwiseArgs(A=p(As[i]), lda=k, Wp=p(Wh), bias=None, R=None, ldr=n, C=p(Cs[i]), ldc=n,
                               M=m, N=n, K=k, relu=0, relu_in=0, sub=1, Wh=p(Wh), a_amax=p(slot)) for i in range(2)]
    def call(i):
      st = streams[i % nstream]
      lib.epos_pointwise_conv_f32(ctypes.byref(args[i % 2]), ctypes.c_void_p(st.cuda_stream))
    # the full kernel once first so that LDS / buffers hold real data for the ablated builds
    phase(vname, call, 2.0 * m * n * k)
    del As, Cs, Wh
phase('idle again', None, 0)
stop[0] = True
