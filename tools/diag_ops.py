#!/usr/bin/env python
"""Diagnostics: per-launch HIP-event times of the network plan and activation
statistics (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import numpy as np
import torch
from epos_amd import model, synthetic, weights

ap = argparse.ArgumentParser()
ap.add_argument('--logits-std', type=float, default=1.0)
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--stats', action='store_true')
args = ap.parse_args()
O, F = 21, 64
ckpt = weights.random_init(num_objs=O, seed=0, logits_std=args.logits_std, randomize_bn=True)
net = model.get_net(ckpt, args.batch, 480, 640, O, F)
img = np.stack([synthetic.image(i, 480, 640) for i in range(args.batch)])
out = net.forward(torch.from_numpy(img).cuda())
torch.cuda.synchronize()
rows = net.time_ops(iters=30, warm=150)
tot = sum(r[1] for r in rows)
agg = {}
for name, ms, fl in rows:
  kind = net.op_kind.get(name, 'other')
  a = agg.setdefault(kind, [0.0, 0.0, 0]); a[0] += ms; a[1] += fl; a[2] += 1
print('total %.3f ms over %d launches' % (tot, len(rows)))
for k, (ms, fl, n) in agg.items():
  print('%-8s n=%3d  %.3f ms  %.1f TFLOP/s' % (k, n, ms, fl / ms / 1e9 if ms else 0))
seen = set()
for name, ms, fl in rows:
  key = name.replace('unit_%s' % name.split('unit_')[-1].split('/')[0], 'unit_N') if 'middle_flow' in name else name
  if 'middle_flow' in name and 'unit_1/' not in name:
    continue
  print('%-95s %8.1f us %7.1f TF' % (name[-95:], ms * 1e3, fl / ms / 1e9 if ms else 0))
if args.stats:
  for nm in ['encoder', 'concat_projection', 'decoder_out']:
    t = getattr(net, nm); print(nm, float(t.abs().mean()), float(t.abs().max()))
  oc = out['pred_obj_conf']; fc = out['pred_frag_conf']
  print('obj conf >0.1 frac per class', [(round(float((oc[..., c] > 0.1).float().mean()), 3)) for c in range(O + 1)])
  print('frag conf max mean', float(fc.max(dim=-1).values.mean()))
