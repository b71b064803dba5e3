"""Full-size runs of the BASELINE.json configurations that are parity-test cases
(not bench lines): C4 (T-LESS-like, 540x720, 30 objects, multi-instance) and C5
(LM-O-like, ResNet-v1-101-beta, batch 8). Checks shapes/finite values of the whole
HIP path and the corr + fitting stages against the oracle chain on the HIP heads."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracle_chain(pipe, store, pred, slots, wants, Ks, seed):
  from oracle import corresp_ref, pnp_ref
  exp = []
  for (im, obj_id), want in zip(slots, wants):
    c = corresp_ref.establish_many_to_many(
        pred['pred_obj_conf'][im], pred['pred_frag_conf'][im],
        pred['pred_frag_loc'][im], [obj_id], store.dp_model['obj_ids'],
        store.frag_centers, store.frag_sizes, 0.25, 0.1, 0.5, True)
    if obj_id not in c:
      continue
    s = (seed * 1000003 + im * 1009 + obj_id) & 0x7fffffffffffffff
    rp, _, rs = pnp_ref.find6DPoses(
        c[obj_id]['coord_2d'], c[obj_id]['coord_3d'], Ks[im],
        params=pnp_ref.default_params(max_model_number=want), seed=s, max_k=4)
    if rp is not None:
      for i in range(rp.shape[0] // 3):
        exp.append((im, obj_id, rp[3 * i:3 * i + 3], rs[i]))
  return exp


def test_c4_tless_like_multi_instance():
  from epos_amd import model, pipeline, synthetic, weights
  O, F, H, W_ = 30, 64, 540, 720
  ckpt = weights.random_init(num_objs=O, seed=2, randomize_bn=True)
  store = synthetic.ModelStore(O, F, seed=1)
  img = synthetic.image(3, H, W_)[None]
  net0 = model.get_net(ckpt, 1, H, W_, O, F)
  net0.forward(torch.from_numpy(img).cuda())
  torch.cuda.synchronize()
  assert (net0.out_h, net0.out_w) == (135, 180)
  synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
  model._NETS.clear()
  pipe = pipeline.EposPipeline(ckpt, 1, H, W_, O, F, store, capacity=1 << 20,
                               max_instances=3)
  targets = [{2: 3, 11: 1, 30: 2}]
  Ks = synthetic.YCBV_K[None]
  poses, _ = pipe.process_batch(torch.from_numpy(img).cuda(), Ks, targets, seed=5)
  pred = {k: v.cpu().numpy() for k, v in pipe.net.forward().items()}
  assert pred['pred_frag_loc'].shape == (1, 135, 180, 30, 64, 3)
  assert all(np.isfinite(v).all() for v in pred.values())
  slots, wants = pipe.make_slots(targets)
  exp = _oracle_chain(pipe, store, pred, slots, wants, Ks, 5)
  assert len(poses) == len(exp)
  for p, (im, obj_id, rp, rs) in zip(poses, exp):
    assert p['obj_id'] == obj_id
    np.testing.assert_allclose(np.hstack([p['R'], p['t']]), rp, atol=1e-9)
    np.testing.assert_allclose(p['score'], rs, rtol=1e-12)


def test_c5_lmo_like_resnet_batch8():
  from epos_amd import model, pipeline, synthetic, weights
  O, F, H, W_, B = 15, 64, 480, 640, 8
  ckpt = weights.random_init('resnet_v1_101_beta', num_objs=O, seed=3,
                             randomize_bn=True)
  store = synthetic.ModelStore(O, F, seed=2)
  mo = model.ModelOptions(model.get_outputs_to_num_channels(O, F),
                          model_variant='resnet_v1_101_beta')
  imgs = np.stack([synthetic.image(i, H, W_) for i in range(B)])
  net0 = model.get_net(ckpt, B, H, W_, O, F, mo)
  net0.forward(torch.from_numpy(imgs).cuda())
  torch.cuda.synchronize()
  synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
  model._NETS.clear()
  del net0
  pipe = pipeline.EposPipeline(ckpt, B, H, W_, O, F, store, capacity=1 << 21,
                               max_instances=1, model_options=mo)
  targets = [synthetic.targets(i, O, 3) for i in range(B)]
  Ks = np.tile(synthetic.YCBV_K, (B, 1, 1))
  poses, rt = pipe.process_batch(torch.from_numpy(imgs).cuda(), Ks, targets,
                                 seed=1, timing=True)
  pred = pipe.net.forward()
  assert tuple(pred['pred_frag_conf'].shape) == (B, 120, 160, O, F)
  assert torch.isfinite(pred['pred_frag_loc']).all()
  assert len(poses) > 0 and rt['total'] > 0
  # images of the batch are independent: image 3 alone gives the same heads
  pipe1 = pipeline.EposPipeline(ckpt, 1, H, W_, O, F, store, capacity=1 << 20,
                                max_instances=1, model_options=mo)
  p1 = pipe1.net.forward(torch.from_numpy(imgs[3:4]).cuda())
  assert torch.equal(p1['pred_obj_conf'][0], pred['pred_obj_conf'][3])


def test_c2_full_size_network_and_pipeline():
  """BASELINE config C2 at its full size (640x480, 21 objects x 64 fragments, 5
  target objects): every head tensor of the HIP network against the torch-CPU
  oracle (fp32, rtol = atol = 3e-4: 65 layers of differently ordered fp32 sums), the
  softmax property sum == 1, and the poses of the device pipeline against the
  oracle chain (numpy correspondences + C RANSAC) run on the HIP heads."""
  from epos_amd import model, pipeline, synthetic, weights
  from oracle import net_ref
  O, F, H, W_ = 21, 64, 480, 640
  ckpt = weights.random_init(num_objs=O, seed=0, randomize_bn=True)
  store = synthetic.ModelStore(O, F, seed=0)
  img = synthetic.image(7, H, W_)[None]
  net0 = model.get_net(ckpt, 1, H, W_, O, F)
  net0.forward(torch.from_numpy(img).cuda())
  torch.cuda.synchronize()
  assert (net0.out_h, net0.out_w) == (120, 160)
  synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
  model._NETS.clear()
  pipe = pipeline.EposPipeline(ckpt, 1, H, W_, O, F, store, capacity=1 << 21,
                               max_instances=1)
  targets = [synthetic.targets(7, O, 5)]
  Ks = synthetic.YCBV_K[None]
  poses, _ = pipe.process_batch(torch.from_numpy(img).cuda(), Ks, targets, seed=3)
  pred = {k: v.cpu().numpy() for k, v in pipe.net.forward().items()}
  ref = net_ref.predict(img, ckpt, num_objs=O, num_frags=F)
  for k in ['pred_obj_conf', 'pred_frag_conf', 'pred_frag_loc']:
    assert pred[k].shape == ref[k].shape, k
    np.testing.assert_allclose(pred[k], ref[k], rtol=3e-4, atol=3e-4, err_msg=k)
  np.testing.assert_allclose(pred['pred_obj_conf'].sum(-1), 1.0, atol=1e-5)
  np.testing.assert_allclose(pred['pred_frag_conf'].sum(-1), 1.0, atol=1e-5)
  conf = np.sort(ref['pred_obj_conf'], axis=-1)
  clear = (conf[..., -1] - conf[..., -2]) > 1e-3
  assert np.array_equal(pred['pred_obj_label'][clear], ref['pred_obj_label'][clear])
  slots, wants = pipe.make_slots(targets)
  exp = _oracle_chain(pipe, store, pred, slots, wants, Ks, 3)
  assert len(poses) == len(exp) and len(poses) >= 3
  for p, (im, obj_id, rp, rs) in zip(poses, exp):
    assert p['obj_id'] == obj_id
    np.testing.assert_allclose(np.hstack([p['R'], p['t']]), rp, atol=1e-9)
    np.testing.assert_allclose(p['score'], rs, rtol=1e-12)
