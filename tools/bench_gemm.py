#!/usr/bin/env python
"""Micro-benchmark of epos_pointwise_conv_f32 on the network's GEMM shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import _lib
lib = _lib.load()
RELU_IN = int(os.environ.get('BENCH_RELU_IN', '0'))
CHECK = os.environ.get('BENCH_CHECK', '1') == '1'
SPLIT = os.environ.get('BENCH_SPLIT', '0') == '1'   # hand the split-packed weights over too
def p(t): return ctypes.c_void_p(t.data_ptr())
shapes = [(4800,728,728,1),(4800,728,728,0),(4800,1024,728,0),(4800,1536,1024,0),(4800,2048,1536,0),(4800,256,2048,0),(4800,256,1280,0),
          (19200,256,304,0),(19200,128,64,0),(19200,256,128,0),(76800,64,288,0),(19200,4032,256,0),(9600,728,728,1),(19200,728,728,1)]
for (m,n,k,res) in shapes:
  A = torch.randn(m,k,device='cuda'); R = torch.randn(m,n,device='cuda'); C = torch.empty(m,n,device='cuda')
  w = (np.random.randn(k,n)/np.sqrt(k)).astype(np.float32)
  total = lib.epos_pack_pointwise_weights(None,k,n,None); dst=np.empty(total,np.float32)
  lib.epos_pack_pointwise_weights(w.ctypes.data_as(ctypes.c_void_p),k,n,dst.ctypes.data_as(ctypes.c_void_p))
  Wp = torch.from_numpy(dst).cuda(); b = torch.zeros((n+127)//128*128,device='cuda')
  Ws = None
  if SPLIT:
    tot = lib.epos_pack_pointwise_weights_split(None,k,n,None); d8=np.empty(tot,np.uint8)
    lib.epos_pack_pointwise_weights_split(w.ctypes.data_as(ctypes.c_void_p),k,n,d8.ctypes.data_as(ctypes.c_void_p))
    Ws = torch.from_numpy(d8).cuda()
  a = _lib.PointwiseArgs(A=p(A),lda=k,Wp=p(Wp),bias=p(b),R=p(R) if res else None,ldr=n,C=p(C),ldc=n,M=m,N=n,K=k,relu=0,relu_in=RELU_IN,sub=1,Ws=p(Ws) if Ws is not None else None)
  call = lambda: lib.epos_pointwise_conv_f32(ctypes.byref(a), None)
  for _ in range(3): call()
  torch.cuda.synchronize()
  err = ''
  if CHECK:
    ref = (torch.relu(A) if RELU_IN else A).double() @ torch.from_numpy(w).cuda().double()
    if res: ref = ref + R.double()
    d = (C.double() - ref).abs().max().item() / ref.abs().max().item()
    err = '  relerr %.1e%s' % (d, '' if d < 1e-5 else '  <-- MISMATCH')
  e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
  for _ in range(int(os.environ.get('BENCH_WARM', '300'))): call()   # let the core clock ramp (2.06 -> 2.4 GHz)
  it=50; e0.record()
  for _ in range(it): call()
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1)/it*1e3
  print('M=%6d N=%5d K=%5d res=%d  %8.1f us  %6.1f TF%s' % (m,n,k,res,us,2*m*n*k/us/1e6,err))
