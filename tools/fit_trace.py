#!/usr/bin/env python
"""Where does the time go INSIDE the fitting kernels? (debug build -DEPOS_FIT_TRACE: the
first workgroup of slot 0 stamps the 100 MHz clock at a few places; C2 bench workload,
serial steps)

    python tools/fit_trace.py build     # here
    python tools/fit_trace.py           # on the GPU box
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epos_amd import build
if len(sys.argv) > 1 and sys.argv[1] == 'build':
  print(build.build_variant('fittrace', ['-DEPOS_FIT_TRACE']))
  sys.exit(0)
os.environ['EPOS_HIP_LIB'] = os.path.join(build.LIB_DIR, 'libepos_hip_fittrace.so')
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
pipe = pipeline.EposPipeline(ckpt, 1, H, W, O, F, store, capacity=1 << 20, max_instances=1,
                             model_options=mo)
lib = _lib.load()
lib.epos_debug_fit_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * (8 * 32))()
cnt = (ctypes.c_int * 8)()
Ks = synthetic.YCBV_K[None]
NAMES = {0: 'ransac_hypotheses (wave 0: entry, before p3p, after p3p, end)',
         1: 'ransac_select_lo (entry, after arg-max, before pass 1, after each pass ..., end)',
         2: 'ransac_refit_accept (entry, before pass 1, after each pass ..., g0 continues, wave 0 '
            'continues, before accept, end)',
         3: 'ransac_gc_scan x2 (entry, window known, after candidate loop, end of tile)',
         4: 'lo_pass of select_lo, repeated: (start, after the point loop, after the wave '
            'butterflies, after lo_combine, after solve + step)'}
for i in range(5):
  img = torch.from_numpy(synthetic.image(i, H, W)[None]).cuda()
  tg = [synthetic.targets(i, O, 5)]
  lib.epos_debug_fit_trace(out, cnt, 1)
  pipe.process_batch(img, Ks, tg, image_ids=[i], seed=i)
  torch.cuda.synchronize()
  lib.epos_debug_fit_trace(out, cnt, 0)
  if i < 2:
    continue
  print('step %d: corr per slot %s' % (i, [int(x) for x in pipe.last_totals[:, 1]]))
  t0 = min(out[k * 32] for k in range(5) if cnt[k])
  for k in range(5):
    ts = [out[k * 32 + j] for j in range(cnt[k])]
    if not ts:
      continue
    print('  %s' % NAMES[k])
    print('    start +%.1f us; deltas (us): %s; total %.1f' % (
        (ts[0] - t0) / 100.0, ' '.join('%.1f' % ((b - a) / 100.0) for a, b in zip(ts, ts[1:])),
        (ts[-1] - ts[0]) / 100.0))
