"""The planted-pose workload of bench.py --planted-poses (epos_amd/synthetic.planted_scene):
what it renders into the heads is a scene the ORACLE chain recovers (CPU), and on the GPU the
scatter kernel + the pipeline's after_net hook deliver exactly that scene to the HIP
correspondence / fitting stages."""
import ctypes

import numpy as np
import pytest

from epos_amd import synthetic


def _apply(scene, P, O, F):
  oc = np.zeros(P * (O + 1), np.float32)
  fc = np.zeros(P * O * F, np.float32)
  fl = np.zeros(P * O * F * 3, np.float32)
  oc[scene['obj'][0]] = scene['obj'][1]
  for (off, val), dst, w in ((scene['frag'], fc, F), (scene['loc'], fl, 3 * F)):
    for o, v in zip(off, val):
      dst[o:o + w] = v
  return oc, fc, fl


@pytest.mark.parametrize('outliers', [0.3, 0.7])
def test_planted_scene_is_recovered_by_the_oracle_chain(outliers):
  from oracle import corresp_ref, pnp_ref
  O, F, h, w = 6, 64, 60, 80
  K = synthetic.YCBV_K.copy()
  K[:2] *= 0.5                                  # a 320 x 240 input: 80 x 60 at stride 4
  store = synthetic.ModelStore(O, F, seed=0)
  tg = {2: 1, 5: 1}
  sc = synthetic.planted_scene(4, store, tg, K, h, w, O, F, outlier_frac=outliers,
                               depth_mm=(300.0, 500.0))
  oc, fc, fl = _apply(sc, h * w, O, F)
  corr = corresp_ref.establish_many_to_many(
      oc.reshape(h, w, O + 1), fc.reshape(h, w, O, F), fl.reshape(h, w, O, F, 3), list(tg),
      store.dp_model['obj_ids'], store.frag_centers, store.frag_sizes, 0.25, 0.1, 0.5, True)
  for obj_id, R, t in sc['poses']:
    masked, n_out = sc['stats'][obj_id]
    c = corr[obj_id]
    assert len(c['coord_2d']) == 2 * masked            # two live fragments per pixel
    est, labels, _ = pnp_ref.find6DPoses(c['coord_2d'], c['coord_3d'], K, seed=obj_id)
    assert est is not None
    rot, tr = synthetic.pose_errors(est[:3, :3], est[:3, 3], R, t)
    assert rot < 1.0 and tr < 5.0, (rot, tr)
    inl = int((np.asarray(labels) >= 0).sum())
    assert 0.8 * 2 * (masked - n_out) <= inl <= 2 * (masked - n_out) + 0.05 * 2 * masked


@pytest.mark.gpu
def test_planted_scene_through_the_hip_pipeline():
  """epos_scatter_blocks_f32 writes the rendering over the network's heads between the
  network and the correspondence stage (EposPipeline.launch(after_net=...)): the correspondence
  counts are the planted ones and every planted pose comes back within 1 degree / 5 mm."""
  import torch
  from epos_amd import _lib, pipeline, weights
  lib = _lib.load()
  O, F, H, W = 6, 64, 240, 320
  K = synthetic.YCBV_K.copy()
  K[:2] *= 0.5
  store = synthetic.ModelStore(O, F, seed=0)
  ckpt = weights.random_init(num_objs=O, seed=1, randomize_bn=True, logits_std=0.6)
  pipe = pipeline.EposPipeline(ckpt, 2, H, W, O, F, store, capacity=1 << 16)
  tgs = [{2: 1, 5: 1}, {1: 2}]
  scenes = [synthetic.planted_scene(10 + b, store, tgs[b], K, pipe.net.out_h, pipe.net.out_w, O,
                                    F, outlier_frac=0.5, image_in_batch=b,
                                    depth_mm=(300.0, 500.0)) for b in range(2)]
  dv = {}
  for key in ('obj', 'frag', 'loc'):
    off = np.concatenate([sc[key][0] for sc in scenes])
    val = np.concatenate([sc[key][1].reshape(len(sc[key][0]), -1) for sc in scenes])
    dv[key] = (torch.from_numpy(off).cuda(), torch.from_numpy(np.ascontiguousarray(val)).cuda(),
               int(val.shape[1]))

  def plant(p):
    st = ctypes.c_void_p(p.stream.cuda_stream)
    for key, name in (('obj', weights.PRED_OBJ_CONF), ('frag', weights.PRED_FRAG_CONF),
                      ('loc', weights.PRED_FRAG_LOC)):
      off, val, width = dv[key]
      _lib.check(lib.epos_scatter_blocks_f32(
          ctypes.c_void_p(p.net.logits[name].data_ptr()), ctypes.c_void_p(off.data_ptr()),
          ctypes.c_void_p(val.data_ptr()), off.numel(), width, st), 'scatter')
  imgs = np.stack([synthetic.image(b, H, W) for b in range(2)])
  poses, _ = pipe.process_batch(torch.from_numpy(imgs).cuda(), np.stack([K, K]), tgs,
                                image_ids=[0, 1], seed=5, after_net=plant)
  # the head buffers hold the rendering
  got = pipe.net.logits[weights.PRED_FRAG_CONF].cpu().numpy().reshape(-1)
  off, val = scenes[1]['frag']
  assert np.array_equal(got[off[3]:off[3] + F], val[3])
  totals = pipe.last_totals                              # [slot, (masked pixels, correspondences)]
  want = [scenes[0]['stats'][2][0], scenes[0]['stats'][5][0], scenes[1]['stats'][1][0]]
  assert [int(x) for x in totals[:, 0]] == want
  assert [int(x) for x in totals[:, 1]] == [2 * x for x in want]
  for b, sc in enumerate(scenes):
    for obj_id, R, t in sc['poses']:
      cand = [synthetic.pose_errors(p['R'], p['t'], R, t) for p in poses
              if p['im_id'] == b and p['obj_id'] == obj_id]
      assert cand and min(c[0] for c in cand) < 1.0 and min(c[1] for c in cand) < 5.0, (b, obj_id, cand)
