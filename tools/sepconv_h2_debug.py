#!/usr/bin/env python
"""Where does the fused fp16-pair separable conv differ from the two launches? (debug aid)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
import test_gpu_h2 as th
shape = [int(x) for x in sys.argv[1:10]] if len(sys.argv) >= 10 else [1, 60, 80, 728, 728, 2, 0, 1, 0]
b, h, w, cin, cout, rate, relu_in, relu_out, res = shape
print('fused state', lib.epos_separable_conv_fused_state(None))
t, m, make = th._sepconv_h2_problem(lib, b, h, w, cin, cout, rate, relu_in, relu_out, res)
T0 = torch.full((m, cin), 3.0, device='cuda'); C0 = torch.zeros(m, cout, device='cuda')
dw, pw, _ = make(T0, C0, None)
_lib.check(lib.epos_depthwise3x3_f32(ctypes.byref(dw), None))
_lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(pw), None))
torch.cuda.synchronize()
sync = torch.zeros(int(lib.epos_separable_conv_sync_words(m)), dtype=torch.int32, device='cuda')
stats = torch.zeros(2, dtype=torch.int32, device='cuda')
T1 = torch.full((m, cin), -7.0, device='cuda'); C1 = torch.empty(m, cout, device='cuda')
_, _, sa = make(T1, C1, sync, stats)
for it in range(2):
  _lib.check(lib.epos_separable_conv_f32(ctypes.byref(sa), None))
  torch.cuda.synchronize()
  a = T1.view(torch.int32).cpu().numpy().reshape(m, cin // 4, 4)
  r = T0.view(torch.int32).cpu().numpy().reshape(m, cin // 4, 4)
  bad = (a != r).any(-1)                       # [m, groups]
  print('launch %d: %d of %d (pixel, group) cells differ; C equal: %s; timeouts %d' % (
      it, bad.sum(), bad.size, bool(torch.equal(C1, C0)), int(stats[0])))
  if bad.any():
    px, gr = np.nonzero(bad)
    print(' pixels: min %d max %d, distinct %d; groups: distinct %s' % (px.min(), px.max(), len(set(px)), sorted(set(gr))[:40]))
    print(' pixel %% 128 histogram (16 bins):', np.histogram(px % 128, bins=16, range=(0, 128))[0])
    print(' group %% 4 histogram:', np.bincount(gr % 4, minlength=4))
    unt = (a == np.float32(-7.0).view(np.int32)).all(-1)
    print(' never written cells:', int(unt.sum()))
    i = 0
    print(' first bad cell', px[i], gr[i], 'got', a[px[i], gr[i]].view(np.float16), 'want', r[px[i], gr[i]].view(np.float16))
    x = px % w; y = (px // w) % h
    print(' x range', x.min(), x.max(), ' y range', y.min(), y.max())
