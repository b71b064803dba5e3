import sys, numpy as np, torch
sys.path.insert(0, '.')
from epos_amd import model, pipeline, synthetic, weights
O, F, H, W = 21, 64, 480, 640
ckpt = weights.random_init('xception_65', num_objs=O, num_frags=F, seed=0, randomize_bn=True)
mo = model.ModelOptions(model.get_outputs_to_num_channels(O, F))
store = synthetic.ModelStore(O, F, seed=0)
net0 = model.get_net(ckpt, 1, H, W, O, F, mo)
net0.forward(torch.from_numpy(synthetic.image(0, H, W)[None]).cuda()); torch.cuda.synchronize()
synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
model._NETS.clear(); del net0
pipe = pipeline.EposPipeline(ckpt, 1, H, W, O, F, store, capacity=1 << 20, max_instances=1, model_options=mo)
for i in range(3):
  img = torch.from_numpy(synthetic.image(i, H, W)[None]).cuda()
  tg = [{o: 1 for o in synthetic.targets(i, O, 5)}]
  poses, _ = pipe.process_batch(img, synthetic.YCBV_K[None], tg, image_ids=[i], seed=i)
  torch.cuda.synchronize()
  tot = pipe.last_totals
  lab = pipe.labels.cpu().numpy()
  base = 0
  for s in range(tot.shape[0]):
    n = int(tot[s, 1]); l = lab[base:base + n]; base += n
    print('image', i, 'slot', s, 'correspondences', n, 'inliers of the accepted pose', int((l >= 0).sum()), 'score', [round(p['score'], 1) for p in poses][s] if s < len(poses) else None)
