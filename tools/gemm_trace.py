#!/usr/bin/env python
"""Per-workgroup phase timeline of the LDS-DMA GEMM (s_memrealtime, 10 ns ticks).

Builds a -DEPOS_GEMM_TRACE copy of the library next to the real one and runs one
shape:  python tools/gemm_trace.py M N K [res]
"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib, build

defs = os.environ.get('TRACE_DEFS', '').split()      # e.g. -DEPOS_ABL_NOBAR (ablations)
tag = ''.join(d.replace('-D', '_') for d in defs)
path = os.path.join(build.LIB_DIR, 'libepos_hip_trace%s.so' % tag)
src = os.path.join(build.CSRC, "pointwise_gemm_dma.hip")
if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
  subprocess.check_call([build.HIPCC] + build.FLAGS + ['-DEPOS_GEMM_TRACE'] + defs +
                        ['-o', path, src, os.path.join(build.CSRC, 'pointwise_gemm.hip'),
                         os.path.join(build.CSRC, 'runtime.hip')])
print('defs:', defs)
lib = ctypes.CDLL(path)
lib.epos_pack_pointwise_weights.restype = ctypes.c_int64
m, n, k = [int(x) for x in sys.argv[1:4]]
res = int(sys.argv[4]) if len(sys.argv) > 4 else 0
def p(t): return ctypes.c_void_p(t.data_ptr())
A = torch.randn(m, k, device='cuda'); R = torch.randn(m, n, device='cuda')
C = torch.empty(m, n, device='cuda')
w = (np.random.randn(k, n) / np.sqrt(k)).astype(np.float32)
total = lib.epos_pack_pointwise_weights(None, k, n, None); dst = np.empty(total, np.float32)
lib.epos_pack_pointwise_weights(w.ctypes.data_as(ctypes.c_void_p), k, n,
                                dst.ctypes.data_as(ctypes.c_void_p))
Wp = torch.from_numpy(dst).cuda(); b = torch.zeros((n + 127) // 128 * 128, device='cuda')
a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Wp), bias=p(b), R=p(R) if res else None, ldr=n,
                       C=p(C), ldc=n, M=m, N=n, K=k, relu=0, relu_in=0, sub=1)
for _ in range(int(os.environ.get('TRACE_WARM', '400'))):
  lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
torch.cuda.synchronize()
nwg = ((m + 63) // 64) * ((n + 127) // 128)
buf = np.zeros(8192 * 8, np.uint64)
lib.epos_debug_read_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
tr = buf.reshape(8192, 8)[:nwg].astype(np.int64)
t0 = tr[:, 0].min()
us = lambda x: x / 100.0
start, pro, loop, epi = us(tr[:, 0] - t0), us(tr[:, 1] - tr[:, 0]), us(tr[:, 2] - tr[:, 1]), us(tr[:, 3] - tr[:, 2])
end = us(tr[:, 3] - t0)
def st(name, v): print('%-10s mean %7.2f  min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us' % (
    name, v.mean(), v.min(), np.median(v), np.percentile(v, 90), v.max()))
mhz = ((tr[:, 6] - tr[:, 5]) / np.maximum(tr[:, 3] - tr[:, 0], 1)).mean() * 100.0
print('M=%d N=%d K=%d res=%d  workgroups %d  kernel span %.2f us  core clock %.0f MHz' % (m, n, k, res, nwg, end.max(), mhz))
st('start', start); st('prologue', pro); st('k-loop', loop); st('epilogue', epi); st('end', end)
hw = tr[:, 4]
cu_key = ((hw >> 32) << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15)   # xcc, se, cu
keys, counts = np.unique(cu_key, return_counts=True)
print('distinct CUs %d; workgroups per CU histogram %s' % (len(keys), dict(zip(*np.unique(counts, return_counts=True)))))
# k-loop time by co-residency
per = dict(zip(keys, counts)); co = np.array([per[x] for x in cu_key])
for c in sorted(set(co)):
  print('  CUs with %d WG: k-loop mean %.2f us, end mean %.2f us' % (c, loop[co == c].mean(), end[co == c].mean()))

try:
  lib.epos_debug_read_trace_units
  ub = np.zeros(512 * 32, np.uint64)
  lib.epos_debug_read_trace_units(ub.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(ub.nbytes))
  ub = ub.reshape(512, 32).astype(np.int64)[:min(nwg, 512)]
  k0 = min(nwg, 512)
  st('pro:args', us(ub[:, 0] - tr[:k0, 0])); st('pro:issue', us(ub[:, 1] - ub[:, 0])); st('pro:wait', us(tr[:k0, 1] - ub[:, 1]))
except AttributeError:
  pass
