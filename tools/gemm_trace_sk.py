#!/usr/bin/env python
"""Per-worker timeline of the persistent stream-K GEMM (s_memrealtime, 10 ns ticks):
python tools/gemm_trace_sk.py M N K [res]   (builds a -DEPOS_GEMM_TRACE library copy)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib, build
defs = os.environ.get('TRACE_DEFS', '').split()
tag = ''.join(d.replace('-D', '_') for d in defs)
path = os.path.join(build.LIB_DIR, 'libepos_hip_trace%s.so' % tag)
src = os.path.join(build.CSRC, "pointwise_gemm_dma.hip")
if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
  subprocess.check_call([build.HIPCC] + build.FLAGS + ['-DEPOS_GEMM_TRACE'] + defs +
                        ['-o', path, src, os.path.join(build.CSRC, 'pointwise_gemm.hip'),
                         os.path.join(build.CSRC, 'runtime.hip')])
lib = ctypes.CDLL(path)
lib.epos_pack_pointwise_weights.restype = ctypes.c_int64
lib.epos_pointwise_workspace_bytes.restype = ctypes.c_int64
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
WS = torch.zeros(int(lib.epos_pointwise_workspace_bytes()), dtype=torch.uint8, device='cuda')
a = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Wp), bias=p(b), R=p(R) if res else None, ldr=n,
                       C=p(C), ldc=n, M=m, N=n, K=k, relu=0, relu_in=0, sub=1)
for _ in range(int(os.environ.get('TRACE_WARM', '400'))):
  lib.epos_pointwise_conv_grouped_sk_f32(ctypes.byref(a), 1, p(WS), None)
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, np.uint64)
lib.epos_debug_read_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
tr = buf.reshape(8192, 8)[:512].astype(np.int64)
tr = tr[tr[:, 3] > 0]
t0 = tr[:, 0].min()
us = lambda x: x / 100.0
def st(name, v): print('%-12s mean %7.2f  min %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f' % (
    name, v.mean(), v.min(), np.median(v), np.percentile(v, 90), v.max()))
print('M=%d N=%d K=%d res=%d  workers %d  kernel span %.2f us' % (m, n, k, res, len(tr), us(tr[:, 3] - t0).max()))
st('start us', us(tr[:, 0] - t0)); st('prologue us', us(tr[:, 1] - tr[:, 0]))
st('total us', us(tr[:, 3] - tr[:, 0])); st('epilogues us', us(tr[:, 2])); st('flagwait us', us(tr[:, 5]))
st('segments', tr[:, 6].astype(float)); st('units', tr[:, 7].astype(float))
st('loop us/unit', us(tr[:, 3] - tr[:, 1] - tr[:, 2]) / tr[:, 7])

hw = tr[:, 4]
cu_key = ((hw >> 32) << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15)
raw = np.arange(len(buf.reshape(8192, 8)[:512]))[buf.reshape(8192, 8)[:512, 3] > 0]
W = 512
x, idx = raw & 7, raw >> 3
L = x * (W // 8) + idx
U = ((m + 63) // 64) * ((n + 127) // 128) * ((k + 31) // 32)
nk = (k + 31) // 32
ub = L * U // W
first_end = (nk - ub % nk) % nk          # units until the first tile boundary
print('co-resident pairs (first 12 CUs): (L, units to first tile end, total us, epilogue us)')
keys = np.unique(cu_key)
for key in keys[:12]:
  sel = np.where(cu_key == key)[0]
  print('  ', [(int(L[i]), int(first_end[i]), float(us(tr[i, 3] - tr[i, 0])), float(us(tr[i, 2]))) for i in sel])
cnt = np.unique(cu_key, return_counts=True)[1]
print('workers per CU histogram', dict(zip(*np.unique(cnt, return_counts=True))))

ub2 = np.zeros(512 * 32, np.uint64)
lib.epos_debug_read_trace_units(ub2.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(ub2.nbytes))
ub2 = ub2.reshape(512, 32).astype(np.int64)
for key in keys[:3]:
  sel = np.where(cu_key == key)[0]
  for i in sel:
    r = int(raw[i]); ts = ub2[r]; ts = ts[ts > 0]
    rel = us(ts - t0)
    print('   L=%d unit end times (us):' % L[i], ' '.join('%.1f' % v for v in rel))
