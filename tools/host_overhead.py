import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from epos_amd import pipeline, synthetic, weights, model
O, F = 21, 64
ckpt = weights.random_init(num_objs=O, seed=0, randomize_bn=True)
store = synthetic.ModelStore(O, F, seed=0)
net0 = model.get_net(ckpt, 1, 480, 640, O, F)
net0.forward(torch.from_numpy(synthetic.image(0, 480, 640)[None]).cuda()); torch.cuda.synchronize()
synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy()); model._NETS.clear(); del net0
pipes = [pipeline.EposPipeline(ckpt, 1, 480, 640, O, F, store, capacity=1 << 21, max_instances=1, instance=j) for j in range(4)]
img = torch.from_numpy(synthetic.image(1, 480, 640)[None]).cuda()
tg = [synthetic.targets(1, O, 5)]; Ks = np.tile(synthetic.YCBV_K, (1, 1, 1))
for p in pipes:
  p.launch(img, Ks, tg, image_ids=[1], seed=0); p.collect()
torch.cuda.synchronize()
tl = tc = 0.0; n = 200
t0 = time.perf_counter()
infl = []
for i in range(n):
  p = pipes[i % 4]
  if len(infl) == 4:
    q = infl.pop(0); a = time.perf_counter(); q._done.synchronize(); b = time.perf_counter(); q.collect(); tc += time.perf_counter() - b
  a = time.perf_counter(); p.launch(img, Ks, tg, image_ids=[1], seed=i); tl += time.perf_counter() - a
  infl.append(p)
while infl:
  q = infl.pop(0); q._done.synchronize(); b = time.perf_counter(); q.collect(); tc += time.perf_counter() - b
tot = time.perf_counter() - t0
print('per step: wall %.3f ms, host in launch() %.3f ms, host in collect() after the wait %.3f ms' % (tot / n * 1e3, tl / n * 1e3, tc / n * 1e3))
