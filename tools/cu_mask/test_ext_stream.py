import ctypes, os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from epos_amd import _lib
lib = _lib.load()
words = (ctypes.c_uint32 * 8)()
for i in range(0, 64): words[i >> 5] |= 1 << (i & 31)
raw = ctypes.c_void_p()
print('create', lib.epos_stream_create_cu_mask(words, 8, ctypes.byref(raw)), raw.value, flush=True)
s = torch.cuda.ExternalStream(raw.value, device=torch.device('cuda:0'))
print('wrapped', s, flush=True)
x = torch.randn(1 << 20, device='cuda')
with torch.cuda.stream(s):
  y = x * 2
s.synchronize()
print('eager ok', float(y.sum()), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
  y2 = x * 3
print('captured', flush=True)
with torch.cuda.stream(s):
  g.replay()
s.synchronize()
print('replay ok', float(y2.sum()), flush=True)
