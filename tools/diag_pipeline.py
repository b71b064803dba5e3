#!/usr/bin/env python
"""Stage timings of the end-to-end pipeline + correspondence statistics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, time
import numpy as np, torch
from epos_amd import pipeline, synthetic, weights
ap = argparse.ArgumentParser()
ap.add_argument('--logits-std', type=float, default=1.0)
ap.add_argument('--randomize-bn', type=int, default=1)
ap.add_argument('--batch', type=int, default=1)
args = ap.parse_args()
O, F = 21, 64
ckpt = weights.random_init(num_objs=O, seed=0, logits_std=args.logits_std, randomize_bn=bool(args.randomize_bn))
store = synthetic.ModelStore(O, F)
B = args.batch
imgs = torch.from_numpy(np.stack([synthetic.image(i, 480, 640) for i in range(B)])).cuda()
from epos_amd import model
net0 = model.get_net(ckpt, B, 480, 640, O, F)
net0.forward(imgs); torch.cuda.synchronize()
synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
model._NETS.clear(); del net0
pipe = pipeline.EposPipeline(ckpt, args.batch, 480, 640, O, F, store, capacity=1 << 21, max_instances=1)
tg = [synthetic.targets(i, O, 5) for i in range(B)]
Ks = np.tile(synthetic.YCBV_K, (B, 1, 1))
for it in range(3):
  poses, rt = pipe.process_batch(imgs, Ks, tg, seed=it, timing=True)
print('stage times (s):', {k: round(v, 5) for k, v in rt.items()}, 'poses', len(poses))
tot = pipe.last_totals
print('masked px / corr per slot:', tot.tolist())
net = pipe.net
for nm in ['encoder', 'decoder_out']:
  t = getattr(net, nm); print(nm, 'mean|x|', float(t.abs().mean()), 'max', float(t.abs().max()))
oc = net.logits['pred_obj_conf']
print('obj conf>0.1 frac:', [round(float((oc[..., c] > 0.1).float().mean()), 3) for c in range(O + 1)])
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for it in range(n):
  pipe.process_batch(imgs, Ks, tg, seed=it)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print('end-to-end %.3f ms/step  %.1f img/s' % (dt * 1e3, B / dt))
