#!/usr/bin/env python
"""How many candidates does a tile of the spatial-coherence sweep stream? (debug build
-DEPOS_GC_STATS of the library; C2 bench workload, a few serial steps)

    python tools/gc_stats.py build     # here
    python tools/gc_stats.py           # on the GPU box
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epos_amd import build
if len(sys.argv) > 1 and sys.argv[1] == 'build':
  print(build.build_variant('gcstats', ['-DEPOS_GC_STATS']))
  sys.exit(0)
os.environ['EPOS_HIP_LIB'] = os.path.join(build.LIB_DIR, 'libepos_hip_gcstats.so')
import numpy as np, torch
from epos_amd import _lib, model, pipeline, synthetic, weights
O, F, H, W = 21, 64, 480, 640
ckpt = weights.random_init('xception_65', num_objs=O, num_frags=F, seed=0, randomize_bn=True)
mo = model.ModelOptions(model.get_outputs_to_num_channels(O, F), model_variant='xception_65')
store = synthetic.ModelStore(O, F, seed=0)
net0 = model.get_net(ckpt, 1, H, W, O, F, mo, device='cuda:0')
net0.forward(torch.from_numpy(synthetic.image(0, H, W)[None]).cuda())
torch.cuda.synchronize()
synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
model._NETS.clear()
pipe = pipeline.EposPipeline(ckpt, 1, H, W, O, F, store, capacity=1 << 21, max_instances=1,
                             model_options=mo)
lib = _lib.load()
lib.epos_debug_gc_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 4)()
Ks = synthetic.YCBV_K[None]
for i in range(4):
  img = torch.from_numpy(synthetic.image(i, H, W)[None]).cuda()
  tg = [synthetic.targets(i, O, 5)]
  lib.epos_debug_gc_stats(out, 1)
  pipe.process_batch(img, Ks, tg, image_ids=[i], seed=i)
  torch.cuda.synchronize()
  lib.epos_debug_gc_stats(out, 0)
  tiles, cand, nsum = out[0], out[1], out[2]
  print('step %d: corr per slot %s; %d tile visits (2 sweeps), mean window %.1f candidates, '
        'mean slot size %.0f' % (i, [int(x) for x in pipe.last_totals[:, 1]], tiles,
                                 cand / max(tiles, 1), nsum / max(tiles, 1)))
