#!/usr/bin/env python
"""Which buffer of the plan first differs when several plans run at once? Builds N plans of
the same network, takes plan 0 alone as the reference, then replays all N graphs concurrently
on N streams and compares EVERY activation buffer (and the absmax slot table) bit for bit, in
plan order. Prints the first differing buffers per plan and round."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from epos_amd import model, weights
O, F, H, W_ = 4, 64, 192, 256
N = int(os.environ.get('NPLANS', 4))
ckpt = weights.random_init(num_objs=O, seed=9, randomize_bn=True, logits_std=0.6)
nets = [model.get_net(ckpt, 1, H, W_, O, F, None, 'cuda:0', j) for j in range(N)]
streams = [torch.cuda.Stream() for _ in range(N)]
rng = np.random.RandomState(3)
frames = [torch.from_numpy(rng.randint(0, 256, (1, H, W_, 3)).astype('f')).cuda() for _ in range(3)]
def bufs(net):
  return [t for t in net._keep if t.dtype == torch.float32 and t.dim() >= 2 and t.numel() > 4096]
net0 = nets[0]
for j, net in enumerate(nets):
  with torch.cuda.stream(streams[j]):
    net.forward(frames[0], use_graph=True)
torch.cuda.synchronize()
ref = {}
for f, img in enumerate(frames):
  with torch.cuda.stream(streams[0]):
    net0.forward(img, use_graph=True)
  torch.cuda.synchronize()
  ref[f] = ([b.clone() for b in bufs(net0)], net0._amax_table.clone())
nb = len(ref[0][0])
print('%d buffers per plan, %d slots' % (nb, net0._n_slots))
expr = {}
for b_i, b in enumerate(bufs(net0)):
  e = net0._exprs.get(id(b))
  expr[b_i] = (e[0][2][:90] if e else '?') + ' ' + str(tuple(b.shape))
bad_total = 0
for rnd in range(30):
  for f, img in enumerate(frames):
    for j, net in enumerate(nets):
      with torch.cuda.stream(streams[j]):
        net.forward(img, use_graph=True)
    torch.cuda.synchronize()
    for j, net in enumerate(nets):
      bl = bufs(net)
      diffs = [i for i in range(nb) if not torch.equal(bl[i], ref[f][0][i])]
      sd = (net._amax_table != ref[f][1]).nonzero().flatten().tolist()
      if diffs or sd:
        bad_total += 1
        if bad_total <= 12:
          i = diffs[0] if diffs else -1
          nd = int((bl[i] != ref[f][0][i]).sum()) if diffs else 0
          mx = float((bl[i] - ref[f][0][i]).abs().max()) if diffs else 0
          print('round %d frame %d plan %d: %d buffers differ, first #%d %s (%d elems, max |d| %.3g); slot words differing: %s'
                % (rnd, f, j, len(diffs), i, expr.get(i), nd, mx, [(w // 64, w % 64) for w in sd[:6]]))
print('plans x rounds with a difference:', bad_total)
