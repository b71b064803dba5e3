#!/usr/bin/env python
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
shapes = [(64*256,128,728),(64*512,128,728),(64*768,128,728),(64*1024,128,728),(64*128,128,728),(64*256,128,2048),(64*256,256,728),(64*75,768,728),(64*75,768,736),(64*64,768,728),(64*86,768,728)]
for (m,n,k) in shapes:
  A = torch.randn(m,k,device='cuda'); C = torch.empty(m,n,device='cuda')
  w = (np.random.randn(k,n)/np.sqrt(k)).astype(np.float32)
  total = lib.epos_pack_pointwise_weights(None,k,n,None); dst=np.empty(total,np.float32)
  lib.epos_pack_pointwise_weights(w.ctypes.data_as(ctypes.c_void_p),k,n,dst.ctypes.data_as(ctypes.c_void_p))
  Wp = torch.from_numpy(dst).cuda(); b = torch.zeros((n+127)//128*128,device='cuda')
  a = _lib.PointwiseArgs(A=p(A),lda=k,Wp=p(Wp),bias=p(b),R=None,ldr=n,C=p(C),ldc=n,M=m,N=n,K=k,relu=0,relu_in=0,sub=1)
  for _ in range(3): lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
  torch.cuda.synchronize()
  e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
  it=20; e0.record()
  for _ in range(it): lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1)/it*1e3
  blocks = ((m+63)//64)*((n+127)//128)
  print('M=%6d N=%5d K=%5d blocks=%5d (%.2f/CU) %8.1f us  %6.1f TF' % (m,n,k,blocks,blocks/256,us,2*m*n*k/us/1e6))
