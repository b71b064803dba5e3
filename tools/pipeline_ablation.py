#!/usr/bin/env python
"""What each kernel family costs END TO END with several steps in flight: the bench
loop (C2, depth 4) with families of ops removed from the network plan before the graph
capture (results are garbage by construction; only the times mean something).

    python tools/pipeline_ablation.py            # full, -dw, -gemm, -fit ...
"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import model, pipeline, synthetic, weights

H, W, O, F, B, DEPTH = 480, 640, 21, 64, 1, int(os.environ.get('DEPTH', '4'))
dev = torch.device('cuda:0')
ckpt = weights.random_init('xception_65', num_objs=O, num_frags=F, seed=0, randomize_bn=True)
store = synthetic.ModelStore(O, F, seed=0)
net0 = model.get_net(ckpt, 1, H, W, O, F, device=dev)
net0.forward(torch.from_numpy(synthetic.image(0, H, W)[None]).to(dev))
torch.cuda.synchronize()
synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
model._NETS.clear(); del net0
pool = []
for j in range(4):
  imgs = np.stack([synthetic.image(j, H, W)])
  pool.append((torch.from_numpy(imgs).to(dev), [synthetic.targets(j, O, 5)], [j]))
Ks = np.tile(synthetic.YCBV_K, (B, 1, 1))

def measure(drop, steps=80, inst0=0):
  pipes = [pipeline.EposPipeline(ckpt, B, H, W, O, F, store, capacity=1 << 21, max_instances=1,
                                 device=dev, use_graph=True, instance=inst0 + j) for j in range(DEPTH)]
  for p in pipes:
    net = p.net
    net.ops = [(n, f) for n, f in net.ops if net.op_kind.get(n) not in drop]
    if 'fit' in drop or 'corr' in drop:     # stub out the C entry points of this pipeline
      class _Stub(object):
        def __init__(self, lib, names):
          self._lib, self._names = lib, names
        def __getattr__(self, k):
          if k in self._names:
            return lambda *a: 0
          return getattr(self._lib, k)
      names = set()
      if 'fit' in drop:
        names.add('epos_find6d_poses_device')
      p.lib = _Stub(p.lib, names)
      if 'corr' in drop:
        p.corr.count = lambda *a, **k: None
        p.corr.fill = lambda *a, **k: None
  def run(count):
    inflight = []
    for i in range(count):
      p = pipes[i % DEPTH]
      if len(inflight) == DEPTH:
        inflight.pop(0).collect()
      imgs, tg, idx = pool[i % 4]
      p.launch(imgs, Ks, tg, image_ids=idx, seed=i)
      inflight.append(p)
    while inflight:
      inflight.pop(0).collect()
  run(DEPTH); run(10)
  torch.cuda.synchronize()
  t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / steps
  del pipes
  model._NETS.clear()
  return dt * 1e3

base = None
CASES = [(), ('dw',), ('gemm',), ('dw', 'other', 'im2col'), ()]
if os.environ.get('ABL_CASES'):
  CASES = [tuple(c.split('+')) if c else () for c in os.environ['ABL_CASES'].split(',')]
for k, drop in enumerate(CASES):
  ms = measure(set(drop), inst0=k * DEPTH)
  base = base or ms
  print('without %-22s %.3f ms/step  (%+.3f)' % ('+'.join(drop) or '(nothing)', ms, ms - base))
