import os, sys
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/epos_amd') else os.getcwd())
import numpy as np, torch
from epos_amd import model, synthetic, weights
O, F = 21, 64
ckpt = weights.random_init(num_objs=O, seed=0, logits_std=1.0, randomize_bn=True)
net = model.get_net(ckpt, 1, 480, 640, O, F)
img = np.stack([synthetic.image(0, 480, 640)])
out = net.forward(torch.from_numpy(img).cuda()); torch.cuda.synchronize()
rows = net.time_ops(iters=30, warm=150)
for name, ms, fl in rows:
  print('%-100s %8.1f' % (name[-100:], ms * 1e3))
