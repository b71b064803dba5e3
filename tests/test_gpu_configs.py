"""Full-size runs of the BASELINE.json configurations that are parity-test cases
(not bench lines): C4 (T-LESS-like, 540x720, 30 objects, multi-instance) and C5
(LM-O-like, ResNet-v1-101-beta, batch 8). Checks shapes/finite values of the whole
HIP path and the corr + fitting stages against the oracle chain on the HIP heads."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_lib():
  from epos_amd import build
  return build.REF_LIB_PATH


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


def _assert_heads_match_oracle(pred, imgs, ckpt, num_objs, num_frags, tol=1e-4, **kw):
  """The three head tensors of the HIP network against oracle/net_ref.predict on the
  same frames, elementwise, at the survey's bar rtol = atol = 1e-4 (the
  confidences are softmax outputs in [0, 1], the coordinates O(1)); labels identical
  wherever the oracle's top-2 confidence gap is clear."""
  from oracle import net_ref
  ref = net_ref.predict(imgs, ckpt, num_objs=num_objs, num_frags=num_frags, **kw)
  for k in ['pred_obj_conf', 'pred_frag_conf', 'pred_frag_loc']:
    a = pred[k] if isinstance(pred[k], np.ndarray) else pred[k].cpu().numpy()
    assert a.shape == ref[k].shape and a.dtype == ref[k].dtype, k
    np.testing.assert_allclose(a, ref[k], rtol=tol, atol=tol, err_msg=k)
  lab = pred['pred_obj_label']
  lab = lab if isinstance(lab, np.ndarray) else lab.cpu().numpy()
  conf = np.sort(ref['pred_obj_conf'], axis=-1)
  clear = (conf[..., -1] - conf[..., -2]) > 1e-3
  assert np.array_equal(lab[clear], ref['pred_obj_label'][clear])
  return ref


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
  _assert_heads_match_oracle(pred, img, ckpt, O, F)
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
  # two images of the batch against the torch-CPU oracle at the full 480x640 size
  sel = [0, 5]
  sub = {k: pred[k][sel].cpu().numpy() for k in pred}
  _assert_heads_match_oracle(sub, imgs[sel], ckpt, O, F,
                             model_variant='resnet_v1_101_beta')
  # and the device poses against the oracle chain on the HIP heads
  predh = {k: v.cpu().numpy() for k, v in pred.items()}
  slots, wants = pipe.make_slots(targets)
  exp = _oracle_chain(pipe, store, predh, slots, wants, Ks, 1)
  assert len(poses) == len(exp)
  for p, (im, obj_id, rp, rs) in zip(poses, exp):
    assert p['obj_id'] == obj_id and p['im_id'] == im
    np.testing.assert_allclose(np.hstack([p['R'], p['t']]), rp, atol=1e-9)


def test_c2_full_size_network_and_pipeline():
  """BASELINE config C2 at its full size (640x480, 21 objects x 64 fragments, 5
  target objects): every head tensor of the HIP network against the torch-CPU
  oracle (fp32, rtol = atol = 1e-4), the raw logits likewise, the
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
  ref = _assert_heads_match_oracle(pred, img, ckpt, O, F)
  np.testing.assert_allclose(pred['pred_obj_conf'].sum(-1), 1.0, atol=1e-5)
  np.testing.assert_allclose(pred['pred_frag_conf'].sum(-1), 1.0, atol=1e-5)
  # raw logits (before the softmax) at the same bar, relative to each head's scale
  pipe.net.run_plan(with_post=False)
  torch.cuda.synchronize()
  for k, v in ref['_logits'].items():
    a = pipe.net.logits[k].cpu().numpy().reshape(v.shape)
    scale = max(1.0, float(np.abs(v).max()))
    np.testing.assert_allclose(a, v, rtol=1e-4, atol=1e-4 * scale, err_msg='logits ' + k)
  slots, wants = pipe.make_slots(targets)
  exp = _oracle_chain(pipe, store, pred, slots, wants, Ks, 3)
  assert len(poses) == len(exp) and len(poses) >= 3
  for p, (im, obj_id, rp, rs) in zip(poses, exp):
    assert p['obj_id'] == obj_id
    np.testing.assert_allclose(np.hstack([p['R'], p['t']]), rp, atol=1e-9)
    np.testing.assert_allclose(p['score'], rs, rtol=1e-12)


def test_c3_per_gpu_shard_batch4():
  """BASELINE config C3 ("YCB-V, batch=32 sharded across 8 MI355X"): the shard ONE GPU
  processes -- 4 images of 640x480, 21 objects, 5 targets per image -- through the
  batched plan: heads against the torch-CPU oracle, poses against the oracle chain,
  and the shard's images bit-identical to the same images processed one by one
  (which is what makes sharding by image legal, scripts/infer.py:712-739)."""
  from epos_amd import dist as edist, model, pipeline, synthetic, weights
  O, F, H, W_, B, G = 21, 64, 480, 640, 4, 8
  ckpt = weights.random_init(num_objs=O, seed=0, randomize_bn=True)
  store = synthetic.ModelStore(O, F, seed=0)
  net0 = model.get_net(ckpt, 1, H, W_, O, F)
  net0.forward(torch.from_numpy(synthetic.image(0, H, W_)[None]).cuda())
  torch.cuda.synchronize()
  synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
  model._NETS.clear()
  del net0
  rank = 5                                  # the sixth GPU's shard of a batch of 32
  b, e = edist.shard_range(B * G, rank, G)
  assert (b, e) == (20, 24)
  idx = list(range(b, e))
  imgs = np.stack([synthetic.image(i, H, W_) for i in idx])
  targets = [synthetic.targets(i, O, 5) for i in idx]
  Ks = np.tile(synthetic.YCBV_K, (B, 1, 1))
  pipe = pipeline.EposPipeline(ckpt, B, H, W_, O, F, store, capacity=1 << 22,
                               max_instances=1)
  poses, _ = pipe.process_batch(torch.from_numpy(imgs).cuda(), Ks, targets,
                                image_ids=idx, seed=9)
  pred = {k: v.cpu().numpy() for k, v in pipe.net.forward().items()}
  assert pred['pred_frag_loc'].shape == (B, 120, 160, O, F, 3)
  _assert_heads_match_oracle(pred, imgs, ckpt, O, F)
  from oracle import corresp_ref, pnp_ref
  slots, wants = pipe.make_slots(targets)
  assert len(slots) == B * 5
  n = 0
  got = {(p['im_id'], p['obj_id']): p for p in poses}
  for (im, obj_id), want in zip(slots, wants):
    c = corresp_ref.establish_many_to_many(
        pred['pred_obj_conf'][im], pred['pred_frag_conf'][im],
        pred['pred_frag_loc'][im], [obj_id], store.dp_model['obj_ids'],
        store.frag_centers, store.frag_sizes, 0.25, 0.1, 0.5, True)
    s = (9 * 1000003 + idx[im] * 1009 + obj_id) & 0x7fffffffffffffff
    rp = None
    if obj_id in c and len(c[obj_id]['coord_2d']) >= 6:
      rp, _, rs = pnp_ref.find6DPoses(
          c[obj_id]['coord_2d'], c[obj_id]['coord_3d'], Ks[im],
          params=pnp_ref.default_params(max_model_number=want), seed=s, max_k=1)
    assert (rp is None) == ((idx[im], obj_id) not in got)
    if rp is not None:
      p = got[(idx[im], obj_id)]
      np.testing.assert_allclose(np.hstack([p['R'], p['t']]), rp[:3], atol=1e-9)
      np.testing.assert_allclose(p['score'], rs[0], rtol=1e-12)
      n += 1
  assert n >= 10
  # image 2 of the shard alone: same head bits, same poses
  pipe1 = pipeline.EposPipeline(ckpt, 1, H, W_, O, F, store, capacity=1 << 21,
                                max_instances=1)
  poses1, _ = pipe1.process_batch(torch.from_numpy(imgs[2:3]).cuda(), Ks[2:3],
                                  targets[2:3], image_ids=idx[2:3], seed=9)
  p1 = pipe1.net.forward()
  for k in ['pred_obj_conf', 'pred_frag_conf', 'pred_frag_loc']:
    assert np.array_equal(p1[k][0].cpu().numpy(), pred[k][2]), k
  ref2 = [p for p in poses if p['im_id'] == idx[2]]
  assert len(poses1) == len(ref2) > 0
  for a, b_ in zip(poses1, ref2):
    assert a['obj_id'] == b_['obj_id'] and a['score'] == b_['score']
    assert np.array_equal(a['R'], b_['R']) and np.array_equal(a['t'], b_['t'])


@pytest.mark.parametrize('tails', ['init', 'heavy_tailed'])
def test_c2_full_size_split_gemm_not_less_accurate_than_fp32_mfma(tmp_path, tails):
  """(tails = 'heavy_tailed', round 6: the same statement on weights.heavy_tailed(checkpoint)
  -- every GEMM matrix has 1 % of its weights at 1e-10 and every fourth column log-normal
  sigma 3, the tails a trained network has; every layer must stay on the fp16-pair kernel.)
  At the full C2 size: the network through the split-operand GEMM (default) and
  through the fp32-MFMA GEMM (EPOS_GEMM_SPLIT=0, read once per process: two
  subprocesses), both against the oracle carried out in fp64 on the same fp32 weights.
  Per head, rms(split - fp64) <= rms(fp32-MFMA - fp64): the precision statement the
  bench line rests on, asserted on the configuration the metric is quoted on. Also
  pins the absolute error of the default path: every raw logit within 1e-4 (relative
  to the head's scale) of the fp64 result."""
  import os
  import subprocess
  import sys
  from oracle import net_ref
  from epos_amd import weights
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  O, F, H, W_ = 21, 64, 480, 640
  script = (
      "import sys, numpy as np, torch\n"
      "sys.path.insert(0, %r)\n"
      "from epos_amd import model, weights, synthetic\n"
      "ckpt = weights.random_init(num_objs=%d, seed=0, randomize_bn=True, logits_std=0.2)\n"
      "if %r == 'heavy_tailed': ckpt = weights.heavy_tailed(ckpt, seed=0)\n"
      "img = synthetic.image(7, %d, %d)[None]\n"
      "net = model.get_net(ckpt, 1, %d, %d, %d, %d)\n"
      "assert not net.h2_refused or not net.use_h2, net.h2_refused\n"
      "assert len(net.h2_layers) >= 70 or not net.use_h2, len(net.h2_layers)\n"
      "net.set_images(torch.from_numpy(img).cuda()); net.run_plan(with_post=False)\n"
      "torch.cuda.synchronize()\n"
      "np.savez(sys.argv[1], **{k: v.cpu().numpy() for k, v in net.logits.items()},\n"
      "         decoder=net.decoder_out.cpu().numpy())\n" % (root, O, tails, H, W_, H, W_, O, F))
  # the two GPU runs (seconds each) go on beside the fp64 oracle of this process (~40 s of CPU)
  procs = {}
  for mode in ('1', '0'):
    path = str(tmp_path / ('logits_%s.npz' % mode))
    procs[mode] = (path, subprocess.Popen(
        [sys.executable, '-c', script, path],
        # mode '0' = the whole plan on the fp32-MFMA kernels: they live in the test build
        env=dict(os.environ, EPOS_GEMM_SPLIT=mode,
                 **({'EPOS_HIP_LIB': _ref_lib()} if mode == '0' else {})),
        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
  from epos_amd import synthetic
  ckpt = weights.random_init(num_objs=O, seed=0, randomize_bn=True, logits_std=0.2)
  if tails == 'heavy_tailed':
    ckpt = weights.heavy_tailed(ckpt, seed=0)
  img = synthetic.image(7, H, W_)[None]
  with torch.no_grad(), net_ref.precision(torch.float64):
    ref, ep = net_ref.logits(img, ckpt, O, F)
  outs = {}
  for mode, (path, pr) in procs.items():
    out, _ = pr.communicate(timeout=900)
    assert pr.returncode == 0, out
    outs[mode] = dict(np.load(path))
  exact = {k: v.permute(0, 2, 3, 1).numpy() for k, v in ref.items()}
  exact['decoder'] = ep['decoder/decoder_conv1'].permute(0, 2, 3, 1).numpy()
  rms = {m: {k: float(np.sqrt(np.mean((outs[m][k].reshape(exact[k].shape)
                                       .astype(np.float64) - exact[k]) ** 2)))
             for k in exact} for m in outs}
  print('C2 full size, rms error vs the fp64 oracle: split', rms['1'], ' fp32 MFMA',
        rms['0'])
  assert not np.array_equal(outs['1']['decoder'], outs['0']['decoder'])
  for k in exact:
    assert rms['1'][k] <= rms['0'][k] * 1.1, (k, rms['1'][k], rms['0'][k])
    scale = max(1.0, float(np.abs(exact[k]).max()))
    err = np.abs(outs['1'][k].reshape(exact[k].shape) - exact[k]).max()
    assert err <= 1e-4 * scale, (k, err, scale)


def test_rccl_initialises_and_gathers_on_this_box():
  """backend='nccl' (= RCCL) on the GPUs that ARE visible: torchrun with one rank per
  device (a single rank on the one-GPU test box), init_process_group through
  epos_amd.dist.init_from_env, one all_gather_into_tensor on device and the pose-record
  gather -- so that the first RCCL call of the project does not happen in the driver's
  scaling run."""
  import os
  import socket
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  n = max(1, torch.cuda.device_count())
  with socket.socket() as sck:
    sck.bind(('127.0.0.1', 0))
    port = sck.getsockname()[1]
  script = (
      "import os, sys\n"
      "sys.path.insert(0, %r)\n"
      "import numpy as np, torch, torch.distributed as dist\n"
      "from epos_amd import dist as ed\n"
      "os.environ['EPOS_FORCE_NCCL'] = '1'\n"
      "rank, world, lr = ed.init_from_env(backend='nccl', force=True)\n"
      "assert dist.is_initialized() and dist.get_backend() == 'nccl'\n"
      "x = torch.full((4,), float(rank + 1), device='cuda:%%d' %% lr)\n"
      "out = torch.empty(4 * world, device=x.device)\n"
      "dist.all_gather_into_tensor(out, x)\n"
      "assert out.cpu().tolist() == [float(r + 1) for r in range(world) for _ in range(4)]\n"
      "poses = [{'scene_id': 1, 'im_id': rank, 'obj_id': 2, 'score': 0.5,\n"
      "          'R': np.eye(3), 't': np.ones((3, 1)), 'time': 0.1}]\n"
      "m = ed.gather_poses(poses, max_records=None)\n"
      "assert [p['im_id'] for p in m] == list(range(world))\n"
      "assert ed.max_over_ranks(float(rank)) == float(world - 1)\n"
      "ed.barrier()\n"
      "if rank == 0: print('RCCL_OK world=%%d' %% world)\n"
      "dist.destroy_process_group()\n" % root)
  path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'epos_rccl_probe.py')
  with open(path, 'w') as f:
    f.write(script)
  out = subprocess.run(
      [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
       str(n), '--master-addr', '127.0.0.1', '--master-port', str(port), path],
      env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'), capture_output=True,
      text=True, timeout=600)
  assert 'RCCL_OK world=%d' % n in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_c1_single_object_frame():
  """BASELINE config C1 ("single 640x480 synthetic RGB frame, 1 object, xc65-f64: the
  plumbing / parity check"): one frame, one object, 64 fragments through the whole device
  path -- heads against the torch-CPU oracle at 1e-4, correspondences bit-exact against
  the numpy oracle, the pose against the C oracle."""
  from epos_amd import model, pipeline, synthetic, weights
  from oracle import corresp_ref, pnp_ref
  O, F, H, W_ = 1, 64, 480, 640
  ckpt = weights.random_init(num_objs=O, seed=4, randomize_bn=True)
  store = synthetic.ModelStore(O, F, seed=3)
  img = synthetic.image(0, H, W_)[None]
  net0 = model.get_net(ckpt, 1, H, W_, O, F)
  net0.forward(torch.from_numpy(img).cuda())
  torch.cuda.synchronize()
  synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
  model._NETS.clear()
  del net0
  pipe = pipeline.EposPipeline(ckpt, 1, H, W_, O, F, store, capacity=1 << 20,
                               max_instances=1)
  Ks = synthetic.YCBV_K[None]
  poses, _ = pipe.process_batch(torch.from_numpy(img).cuda(), Ks, [{1: 1}], seed=2)
  pred = {k: v.cpu().numpy() for k, v in pipe.net.forward().items()}
  assert pred['pred_obj_conf'].shape == (1, 120, 160, 2)
  assert pred['pred_frag_loc'].shape == (1, 120, 160, 1, 64, 3)
  _assert_heads_match_oracle(pred, img, ckpt, O, F)
  # the operator form gives the oracle's correspondences bit for bit
  from epos_amd import corresp
  got = corresp.establish_many_to_many(pred['pred_obj_conf'][0], pred['pred_frag_conf'][0],
                                       pred['pred_frag_loc'][0], [1], store, 0.25, 0.1, 0.5,
                                       False, True)
  ref = corresp_ref.establish_many_to_many(
      pred['pred_obj_conf'][0], pred['pred_frag_conf'][0], pred['pred_frag_loc'][0], [1],
      store.dp_model['obj_ids'], store.frag_centers, store.frag_sizes, 0.25, 0.1, 0.5, True)
  assert set(got) == set(ref) == {1}
  for k in ref[1]:
    assert got[1][k].dtype == ref[1][k].dtype and np.array_equal(got[1][k], ref[1][k]), k
  s = (2 * 1000003 + 0 * 1009 + 1) & 0x7fffffffffffffff
  rp, _, rs = pnp_ref.find6DPoses(ref[1]['coord_2d'], ref[1]['coord_3d'], Ks[0], seed=s,
                                  max_k=1)
  assert (rp is None) == (len(poses) == 0)
  if rp is not None:
    np.testing.assert_allclose(np.hstack([poses[0]['R'], poses[0]['t']]), rp[:3], atol=1e-9)
    np.testing.assert_allclose(poses[0]['score'], rs[0], rtol=1e-12)
