"""End-to-end GPU test: network -> correspondences -> PnP-RANSAC in HBM vs the
same stages chained on the CPU oracle from the SAME head tensors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class Store(object):
  def __init__(self, num_objs, num_frags, seed=0):
    rng = np.random.RandomState(seed)
    self.dp_model = {'obj_ids': list(range(1, num_objs + 1))}
    self.frag_centers = {o: rng.uniform(-80, 80, (num_frags, 3))
                         for o in self.dp_model['obj_ids']}
    self.frag_sizes = {o: rng.uniform(5, 40, num_frags)
                       for o in self.dp_model['obj_ids']}


def test_pipeline_matches_oracle_chain():
  from epos_amd import pipeline, weights
  from oracle import corresp_ref, pnp_ref
  O, F, B, H, W_ = 3, 64, 2, 96, 128
  ckpt = weights.random_init(num_objs=O, seed=5, randomize_bn=True, logits_std=0.6)
  store = Store(O, F)
  pipe = pipeline.EposPipeline(ckpt, B, H, W_, O, F, store, capacity=1 << 18)
  img = np.random.RandomState(1).randint(0, 256, (B, H, W_, 3)).astype('f')
  Ks = np.tile(np.array([[300., 0, 64], [0, 300., 48], [0, 0, 1]]), (B, 1, 1))
  targets = [{1: 1, 3: 1}, {2: 1}]
  poses, times = pipe.process_batch(torch.from_numpy(img).cuda(), Ks, targets,
                                    seed=7, timing=True)
  assert set(times) == {'prediction', 'establish_corr', 'fitting', 'total'}
  # The oracle chain on the head tensors the HIP network produced.
  pred = {k: v.cpu().numpy() for k, v in pipe.net.forward().items()}
  slots, wants = pipe.make_slots(targets)
  exp = []
  for (im, obj_id), want in zip(slots, wants):
    c = corresp_ref.establish_many_to_many(
        pred['pred_obj_conf'][im], pred['pred_frag_conf'][im],
        pred['pred_frag_loc'][im], [obj_id], store.dp_model['obj_ids'],
        store.frag_centers, store.frag_sizes, 0.25, 0.1, 0.5, True)
    if obj_id not in c:
      continue
    seed = (7 * 1000003 + im * 1009 + obj_id) & 0x7fffffffffffffff
    rp, rl, rs = pnp_ref.find6DPoses(
        c[obj_id]['coord_2d'], c[obj_id]['coord_3d'], Ks[im],
        params=pnp_ref.default_params(max_model_number=want), seed=seed)
    if rp is not None:
      exp.append((im, obj_id, rp, rs))
  assert len(poses) == len(exp)
  for p, (im, obj_id, rp, rs) in zip(poses, exp):
    assert (p['im_id'], p['obj_id']) == (im, obj_id)
    np.testing.assert_allclose(np.hstack([p['R'], p['t']]), rp[:3], atol=1e-9)
    np.testing.assert_allclose(p['score'], rs[0], rtol=1e-12)


def test_pipeline_opencv_method_matches_its_oracle_chain():
  """fitting_method='opencv_ransac' in the fused device pipeline: per slot, the pose of
  cv2.solvePnPRansac(EPNP) as oracle/epnp_ref.c states it, on the correspondences the oracle
  extracts from the same head tensors; slots below 6 correspondences are skipped
  (infer.py:420-422), score 0.0, one pose per object even when two instances are asked."""
  from epos_amd import pipeline, weights
  from oracle import corresp_ref, epnp_ref
  O, F, B, H, W_ = 3, 64, 2, 96, 128
  ckpt = weights.random_init(num_objs=O, seed=5, randomize_bn=True, logits_std=0.6)
  store = Store(O, F)
  pipe = pipeline.EposPipeline(ckpt, B, H, W_, O, F, store, capacity=1 << 18,
                               fitting_method='opencv_ransac')
  img = np.random.RandomState(1).randint(0, 256, (B, H, W_, 3)).astype('f')
  Ks = np.tile(np.array([[300., 0, 64], [0, 300., 48], [0, 0, 1]]), (B, 1, 1))
  targets = [{1: 2, 3: 1}, {2: 1}]
  poses, _ = pipe.process_batch(torch.from_numpy(img).cuda(), Ks, targets, seed=7)
  pred = {k: v.cpu().numpy() for k, v in pipe.net.forward().items()}
  slots, wants = pipe.make_slots(targets)
  exp = []
  for (im, obj_id), want in zip(slots, wants):
    c = corresp_ref.establish_many_to_many(
        pred['pred_obj_conf'][im], pred['pred_frag_conf'][im],
        pred['pred_frag_loc'][im], [obj_id], store.dp_model['obj_ids'],
        store.frag_centers, store.frag_sizes, 0.25, 0.1, 0.5, True)
    if obj_id not in c or c[obj_id]['coord_2d'].shape[0] < 6:
      continue
    ok, P, mask, info = epnp_ref.solvePnPRansac(c[obj_id]['coord_3d'], c[obj_id]['coord_2d'],
                                                Ks[im], 400, 4.0, 0.99)
    if ok:
      exp.append((im, obj_id, P))
  assert len(exp) >= 1
  assert len(poses) == len(exp)
  for p, (im, obj_id, P) in zip(poses, exp):
    assert (p['im_id'], p['obj_id']) == (im, obj_id)
    np.testing.assert_array_equal(np.hstack([p['R'], p['t']]), P)
    assert p['score'] == 0.0
  with pytest.raises(ValueError):
    pipeline.EposPipeline(ckpt, B, H, W_, O, F, store, fitting_method='nope')


def test_sparse_heads_equal_dense_on_target_objects():
  """Sparse-head mode evaluates the fragment heads only for the target objects;
  those channels and the resulting poses must be bit-identical to the dense run."""
  from epos_amd import model, pipeline, synthetic, weights
  O, F, B, H, W_ = 6, 64, 2, 96, 128
  ckpt = weights.random_init(num_objs=O, seed=8, randomize_bn=True)
  store = Store(O, F)
  img = np.stack([synthetic.image(i, H, W_) for i in range(B)])
  net0 = model.get_net(ckpt, B, H, W_, O, F)
  net0.forward(torch.from_numpy(img).cuda())
  torch.cuda.synchronize()
  synthetic.calibrate_logits(ckpt, net0.decoder_out[0].cpu().numpy())
  model._NETS.clear()
  Ks = np.tile(np.array([[300., 0, 64], [0, 300., 48], [0, 0, 1]]), (B, 1, 1))
  targets = [{1: 1, 4: 2}, {6: 1}]
  dense = pipeline.EposPipeline(ckpt, B, H, W_, O, F, store, capacity=1 << 18,
                                instance=0)
  sparse = pipeline.EposPipeline(ckpt, B, H, W_, O, F, store, capacity=1 << 18,
                                 instance=1, sparse_heads=True)
  x = torch.from_numpy(img).cuda()
  pd, _ = dense.process_batch(x, Ks, targets, seed=2)
  ps, _ = sparse.process_batch(x, Ks, targets, seed=2, timing=True)
  assert len(pd) == len(ps) and len(pd) > 0
  for a, b in zip(pd, ps):
    assert a['obj_id'] == b['obj_id'] and a['score'] == b['score']
    assert np.array_equal(a['R'], b['R']) and np.array_equal(a['t'], b['t'])
  dc = dense.net.logits['pred_frag_conf'].view(B, -1, O, F)
  sc = sparse.net.logits['pred_frag_conf'].view(B, -1, O, F)
  dl = dense.net.logits['pred_frag_loc'].view(B, -1, O, F * 3)
  sl = sparse.net.logits['pred_frag_loc'].view(B, -1, O, F * 3)
  for im, t in enumerate(targets):
    for obj in t:
      assert torch.equal(dc[im, :, obj - 1], sc[im, :, obj - 1])
      assert torch.equal(dl[im, :, obj - 1], sl[im, :, obj - 1])
  assert torch.equal(dense.net.logits['pred_obj_conf'],
                     sparse.net.logits['pred_obj_conf'])


def test_concurrent_pipelines_are_deterministic():
  """Four plans in flight on four HIP streams (what bench.py does): the LDS-DMA
  rings, counted waits and grouped launches of one plan must not be disturbed by
  the kernels of the others. Every plan processes the same frames in every round;
  head tensors and poses have to repeat bit for bit, round after round, and agree
  between the plans (same weights, same input)."""
  from epos_amd import pipeline, weights
  O, F, B, H, W_ = 4, 64, 1, 192, 256
  ckpt = weights.random_init(num_objs=O, seed=9, randomize_bn=True, logits_std=0.6)
  store = Store(O, F)
  depth = 4
  pipes = [pipeline.EposPipeline(ckpt, B, H, W_, O, F, store, capacity=1 << 19,
                                 instance=j) for j in range(depth)]
  rng = np.random.RandomState(3)
  frames = [torch.from_numpy(rng.randint(0, 256, (B, H, W_, 3)).astype('f')).cuda()
            for _ in range(3)]
  Ks = np.tile(np.array([[400., 0, 128], [0, 400., 96], [0, 0, 1]]), (B, 1, 1))
  targets = [{1: 1, 2: 1, 4: 1}]
  ref = {}
  for rnd in range(6):
    for f, img in enumerate(frames):
      for p in pipes:                          # all four in flight on the same frame
        p.launch(img, Ks, targets, seed=11)
      outs = []
      for p in pipes:
        poses, _ = p.collect()
        logits = p.net.logits['pred_frag_loc'].clone()
        outs.append((poses, logits))
      for poses, logits in outs:
        key = f
        if key not in ref:
          ref[key] = (poses, logits.cpu())
        rp, rl = ref[key]
        assert torch.equal(logits.cpu(), rl)
        assert len(poses) == len(rp)
        for a, b in zip(poses, rp):
          assert a['obj_id'] == b['obj_id'] and a['score'] == b['score']
          assert np.array_equal(a['R'], b['R']) and np.array_equal(a['t'], b['t'])


def test_bench_multi_rank_flow_on_one_gpu():
  """bench.py under torchrun with two ranks mapped onto the one GPU of the test box
  (EPOS_FORCE_DEVICE=0) and gloo for the pose-record all_gather / barriers
  (EPOS_DIST_BACKEND=gloo: RCCL refuses two ranks on one device): the N > 1 code path
  -- rank-local plans, per-step gather, max-over-ranks timing, rank-0 JSON line --
  end to end on real kernels."""
  import json
  import os
  import socket
  import subprocess
  import sys
  with socket.socket() as sck:
    sck.bind(('127.0.0.1', 0))
    port = sck.getsockname()[1]
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, EPOS_DIST_BACKEND='gloo', EPOS_FORCE_DEVICE='0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
         str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4',
         '--warmup', '1', '--no-cpu-baseline', '--no-roofline']
  out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True,
                       timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
  assert len(lines) == 1                      # rank 0 only
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['steps'] == 4 and d['scaling'] == 'weak'
  # N > 1 defaults to the per-GPU shard of config C3 (4 images per GPU and step)
  assert d['config']['global_batch'] == 8 and d['value'] > 0
  assert d['config']['workload'].startswith('C3 per-GPU shard')
  assert d['config']['rccl_ranks_seen'] == 2 and d['config']['dist_backend'] == 'gloo'
  assert d['serial_depth1']['images_per_sec'] > 0
  assert d['config']['poses_per_step'] > 0


def test_capacity_overflow_is_reported_and_contained():
  """More correspondences than the pooled arrays hold: the fill kernel raises the
  overflow flag, the fitting stage treats every slot that would reach beyond the
  arrays as empty (no out-of-range row is read or written on the device), the host
  raises EposError -- and the same process keeps working afterwards."""
  from epos_amd import _lib, model, pipeline, synthetic, weights
  O, F, H, W_ = 3, 64, 96, 128
  ckpt = weights.random_init(num_objs=O, seed=1, randomize_bn=True, logits_std=0.6)
  store = synthetic.ModelStore(O, F, seed=0)
  img = torch.from_numpy(synthetic.image(0, H, W_)[None]).cuda()
  K = np.array([[300., 0, 64], [0, 300., 48], [0, 0, 1]])[None]
  targets = [{1: 1, 2: 1, 3: 1}]
  big = pipeline.EposPipeline(ckpt, 1, H, W_, O, F, store, capacity=1 << 16)
  poses_ok, _ = big.process_batch(img, K, targets, seed=3)
  total = int(big.last_totals[:, 1].sum())
  assert total > 64 and len(poses_ok) > 0
  # guard words behind the (too small) pooled arrays must survive the overflowing run
  small = pipeline.EposPipeline(ckpt, 1, H, W_, O, F, store, capacity=total // 2,
                                instance=1)
  guard = torch.full((4096,), 12345, dtype=torch.int32, device='cuda')
  with pytest.raises(_lib.EposError):
    small.process_batch(img, K, targets, seed=3)
  torch.cuda.synchronize()
  assert int((guard != 12345).sum()) == 0
  again, _ = big.process_batch(img, K, targets, seed=3)
  assert len(again) == len(poses_ok)
  for a, b in zip(again, poses_ok):
    assert np.array_equal(a['R'], b['R']) and np.array_equal(a['t'], b['t'])


def test_object_ids_without_channels_are_rejected():
  """A model store that lists more objects than the network has channels (an LM store
  with an LM-O checkpoint): ValueError on the host instead of out-of-range channel
  reads on the device (the reference raises IndexError at corresp.py:46)."""
  from epos_amd import corresp, pipeline, synthetic, weights
  O, F, H, W_ = 2, 64, 64, 64
  ckpt = weights.random_init(num_objs=O, seed=1)
  store = synthetic.ModelStore(O + 1, F, seed=0)          # lists object 3 too
  pipe = pipeline.EposPipeline(ckpt, 1, H, W_, O, F, store, capacity=1 << 12)
  with pytest.raises(ValueError):
    pipe.make_slots([{1: 1, 3: 1}])
  slots, _ = pipe.make_slots([{1: 1, 2: 1}])             # targets inside: fine
  assert slots == [(0, 1), (0, 2)]
  with pytest.raises(ValueError):
    pipe.make_slots([{}], task_type=pipeline.DETECTION)  # every store object
  z = np.zeros
  with pytest.raises(ValueError):
    corresp.establish_many_to_many(z((16, 16, O + 1), 'f'), z((16, 16, O, F), 'f'),
                                   z((16, 16, O, F, 3), 'f'), [3], store, 0.25, 0.1,
                                   0.5, False, True)


def test_two_batches_enqueued_per_pipeline_give_the_poses_of_one():
  """EposPipeline(queue=2): launch() accepts a second batch before the first is collected (the
  stream then never waits for the host between two batches); the batches run in stream order on
  the same device buffers and only the host-side staging exists twice. Poses, scores and stage
  timers' keys are those of the queue=1 pipeline, batch by batch; a third launch raises."""
  from epos_amd import _lib, pipeline, weights
  O, F, B, H, W_ = 3, 64, 1, 96, 128
  ckpt = weights.random_init(num_objs=O, seed=5, randomize_bn=True, logits_std=0.6)
  store = Store(O, F)
  Ks = np.tile(np.array([[300., 0, 64], [0, 300., 48], [0, 0, 1]]), (B, 1, 1))
  frames = [torch.from_numpy(np.random.RandomState(10 + i).randint(0, 256, (B, H, W_, 3)).astype(
      'f')).cuda() for i in range(5)]
  targets = [[{1: 1, 3: 1}], [{2: 1}], [{1: 1, 2: 1, 3: 1}], [{3: 1}], [{1: 1}]]
  one = pipeline.EposPipeline(ckpt, B, H, W_, O, F, store, capacity=1 << 18)
  exp = [one.process_batch(frames[i], Ks, targets[i], image_ids=[i], seed=3, timing=True)
         for i in range(5)]
  two = pipeline.EposPipeline(ckpt, B, H, W_, O, F, store, capacity=1 << 18, queue=2)
  got, n_in = [], 0
  for i in range(5):
    if n_in == 2:
      got.append(two.collect()); n_in -= 1
    two.launch(frames[i], Ks, targets[i], image_ids=[i], seed=3, timing=True)
    n_in += 1
    if i == 1:
      with pytest.raises(_lib.EposError):
        two.launch(frames[i], Ks, targets[i], image_ids=[i], seed=3)
  while n_in:
    got.append(two.collect()); n_in -= 1
  with pytest.raises(_lib.EposError):
    two.collect()
  assert len(got) == 5 and sum(len(p) for p, _ in exp) >= 3
  for (pe, te), (pg, tg) in zip(exp, got):
    assert set(te) == set(tg)
    assert len(pe) == len(pg)
    for a, b in zip(pe, pg):
      assert (a['im_id'], a['obj_id']) == (b['im_id'], b['obj_id'])
      assert np.array_equal(a['R'], b['R']) and np.array_equal(a['t'], b['t'])
      assert a['score'] == b['score']
