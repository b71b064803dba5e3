#!/usr/bin/env python
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
m,n,k = [int(x) for x in sys.argv[1:4]]; it = int(sys.argv[4]) if len(sys.argv)>4 else 20
A = torch.randn(m,k,device='cuda'); C = torch.empty(m,n,device='cuda')
w = (np.random.randn(k,n)/np.sqrt(k)).astype(np.float32)
total = lib.epos_pack_pointwise_weights(None,k,n,None); dst=np.empty(total,np.float32)
lib.epos_pack_pointwise_weights(w.ctypes.data_as(ctypes.c_void_p),k,n,dst.ctypes.data_as(ctypes.c_void_p))
Wp = torch.from_numpy(dst).cuda(); b = torch.zeros((n+127)//128*128,device='cuda')
Ws = None
if os.environ.get('BENCH_SPLIT', '0') == '1':      # split-operand kernel
  tot = lib.epos_pack_pointwise_weights_split(None,k,n,None); d8 = np.empty(tot,np.uint8)
  lib.epos_pack_pointwise_weights_split(w.ctypes.data_as(ctypes.c_void_p),k,n,d8.ctypes.data_as(ctypes.c_void_p))
  Ws = torch.from_numpy(d8).cuda()
a = _lib.PointwiseArgs(A=p(A),lda=k,Wp=p(Wp),bias=p(b),R=None,ldr=n,C=p(C),ldc=n,M=m,N=n,K=k,relu=0,relu_in=0,sub=1,Ws=p(Ws) if Ws is not None else None)
for _ in range(it): lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
torch.cuda.synchronize()
