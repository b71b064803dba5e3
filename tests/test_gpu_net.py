"""GPU parity of the whole network plan against the torch-CPU oracle
(oracle/net_ref.py), fp32: logits within rtol = atol = 2e-4 of the oracle's
(summation order differs: BN folded, MFMA k-pairing), predictions likewise, object
labels identical except at near-ties."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_lib():
  from epos_amd import build
  return build.REF_LIB_PATH


@pytest.mark.parametrize('h,w,num_objs,batch', [(96, 128, 2, 1), (65, 97, 1, 2)])
def test_net_matches_oracle(h, w, num_objs, batch):
  from epos_amd import model, weights
  from oracle import net_ref
  ckpt = weights.random_init(num_objs=num_objs, seed=3, randomize_bn=True,
                             logits_std=0.2)
  img = np.random.RandomState(0).randint(0, 256, (batch, h, w, 3)).astype('f')
  ref = net_ref.predict(img, ckpt, num_objs=num_objs, num_frags=64)
  mo = model.ModelOptions(model.get_outputs_to_num_channels(num_objs, 64))
  net = model.get_net(ckpt, batch, h, w, num_objs, 64, mo)
  out = net.forward(torch.from_numpy(img).cuda())
  torch.cuda.synchronize()
  ep = ref['_end_points']

  def nhwc(t):
    return t.permute(0, 2, 3, 1).numpy()
  # intermediate check points (plan buffers) and the heads: the bar of the full-size
  # configuration tests (tests/test_gpu_configs.py), rtol = 1e-4 with an absolute term of
  # 1e-4 of the tensor's scale (two fp32 evaluations of a 65-layer network differ by a few
  # 1e-6 of the scale; the intermediates of a random-init network reach 1e1 .. 1e2)
  def close(a, b, what):
    scale = max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-4 * scale, err_msg=what)
  close(net.encoder.cpu().numpy(), nhwc(ep['encoder']), 'encoder')
  close(net.aspp_concat.cpu().numpy(), nhwc(ep['aspp_concat']), 'aspp_concat')
  close(net.decoder_concat.cpu().numpy(), nhwc(ep['decoder_concat']), 'decoder_concat')
  close(net.decoder_out.cpu().numpy(), nhwc(ep['decoder/decoder_conv1']), 'decoder_out')
  for k in ['pred_obj_conf', 'pred_frag_conf', 'pred_frag_loc']:
    a = out[k].cpu().numpy()
    assert a.shape == ref[k].shape and a.dtype == ref[k].dtype, k
    np.testing.assert_allclose(a, ref[k], rtol=1e-4, atol=1e-4, err_msg=k)
  lab = out['pred_obj_label'].cpu().numpy()
  assert lab.dtype == np.int64 and lab.shape == ref['pred_obj_label'].shape
  conf = np.sort(ref['pred_obj_conf'], axis=-1)
  clear = (conf[..., -1] - conf[..., -2]) > 1e-3
  assert np.array_equal(lab[clear], ref['pred_obj_label'][clear])
  # graph replay gives the same bits as eager
  a0 = out['pred_frag_loc'].clone()
  out2 = net.forward(torch.from_numpy(img).cuda(), use_graph=True)
  torch.cuda.synchronize()
  assert torch.equal(a0, out2['pred_frag_loc'])


def test_predict_operator_api():
  from epos_amd import model, weights
  ckpt = weights.random_init(num_objs=1, seed=0)
  img = np.zeros((1, 64, 64, 3), np.float32)
  mo = model.ModelOptions(model.get_outputs_to_num_channels(1, 64))
  out = model.predict(img, mo, ckpt, num_objs=1, num_frags=64)
  assert set(out) == {'pred_obj_conf', 'pred_obj_label', 'pred_frag_conf',
                      'pred_frag_loc'}
  assert tuple(out['pred_frag_loc'].shape) == (1, 16, 16, 1, 64, 3)
  with pytest.raises(NotImplementedError):
    model.predict(img, mo, ckpt, upsample_logits=True, num_objs=1, num_frags=64)


def test_resnet_v1_101_beta_matches_oracle():
  """BASELINE config C5 backbone (net_resnet_v1_beta.py) at a reduced size."""
  from epos_amd import model, weights
  from oracle import net_ref
  num_objs, h, w = 2, 96, 128
  ckpt = weights.random_init('resnet_v1_101_beta', num_objs=num_objs, seed=4,
                             randomize_bn=True, logits_std=0.2)
  img = np.random.RandomState(2).randint(0, 256, (1, h, w, 3)).astype('f')
  ref = net_ref.predict(img, ckpt, num_objs=num_objs, num_frags=64,
                        model_variant='resnet_v1_101_beta')
  mo = model.ModelOptions(model.get_outputs_to_num_channels(num_objs, 64),
                          model_variant='resnet_v1_101_beta')
  net = model.get_net(ckpt, 1, h, w, num_objs, 64, mo)
  out = net.forward(torch.from_numpy(img).cuda())
  torch.cuda.synchronize()
  enc = ref['_end_points']['encoder'].permute(0, 2, 3, 1).numpy()
  scale = float(np.abs(enc).max())
  np.testing.assert_allclose(net.encoder.cpu().numpy(), enc, rtol=1e-4,
                             atol=1e-4 * max(scale, 1.0))
  for k in ['pred_obj_conf', 'pred_frag_conf', 'pred_frag_loc']:
    np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=1e-4, atol=1e-4,
                               err_msg=k)


def test_maxpool_subsample_add_relu():
  import ctypes
  from epos_amd import _lib
  from oracle import net_ref
  lib = _lib.load()
  rng = np.random.RandomState(0)

  def p(t):
    return ctypes.c_void_p(t.data_ptr())
  for (hi, wi) in [(12, 16), (11, 15)]:
    x = rng.standard_normal((2, hi, wi, 8)).astype('f')
    X = torch.from_numpy(x).cuda()
    ref = net_ref.max_pool_3x3_s2_same(torch.from_numpy(x).permute(0, 3, 1, 2))
    ref = ref.permute(0, 2, 3, 1).numpy()
    Y = torch.zeros(ref.shape, device='cuda')
    _lib.check(lib.epos_maxpool3x3_s2_f32(p(X), 8, p(Y), 8, 2, hi, wi, 8, None))
    assert np.array_equal(Y.cpu().numpy(), ref)
    sub = x[:, ::2, ::2]
    Z = torch.zeros(sub.shape, device='cuda')
    _lib.check(lib.epos_subsample_f32(p(X), 8, p(Z), 8, 2, hi, wi, 8, 2, None))
    assert np.array_equal(Z.cpu().numpy(), sub)
  a = rng.standard_normal(1024).astype('f'); b = rng.standard_normal(1024).astype('f')
  A, Bt = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
  Yt = torch.zeros(1024, device='cuda')
  _lib.check(lib.epos_add_relu_f32(p(A), p(Bt), p(Yt), 1024, None))
  assert np.array_equal(Yt.cpu().numpy(), np.maximum(a + b, 0))


def test_split_gemm_network_is_not_less_accurate_than_fp32_mfma(tmp_path):
  """The whole network through the split-operand GEMM (default) and through the
  fp32-MFMA GEMM (EPOS_GEMM_SPLIT=0, read once per process: two subprocesses), both
  against the oracle carried out in fp64 on the same fp32 weights: the split path's
  logits are not further from the exact result than the fp32-MFMA path's."""
  import os
  import subprocess
  import sys
  from oracle import net_ref
  from epos_amd import weights
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  h, w, num_objs = 96, 128, 2
  script = (
      "import sys, numpy as np, torch\n"
      "sys.path.insert(0, %r)\n"
      "from epos_amd import model, weights\n"
      "ckpt = weights.random_init(num_objs=%d, seed=3, randomize_bn=True, logits_std=0.2)\n"
      "img = np.random.RandomState(0).randint(0, 256, (1, %d, %d, 3)).astype('f')\n"
      "mo = model.ModelOptions(model.get_outputs_to_num_channels(%d, 64))\n"
      "net = model.get_net(ckpt, 1, %d, %d, %d, 64, mo)\n"
      "net.forward(torch.from_numpy(img).cuda()); torch.cuda.synchronize()\n"
      "np.savez(sys.argv[1], **{k: v.cpu().numpy() for k, v in net.logits.items()},\n"
      "         decoder=net.decoder_out.cpu().numpy())\n" % (root, num_objs, h, w, num_objs,
                                                           h, w, num_objs))
  procs = {}                               # the two GPU runs side by side, beside the oracle
  for mode in ('1', '0'):
    path = str(tmp_path / ('logits_%s.npz' % mode))
    procs[mode] = (path, subprocess.Popen(
        [sys.executable, '-c', script, path],
        # mode '0' = the whole plan on the fp32-MFMA kernels: they live in the test build
        env=dict(os.environ, EPOS_GEMM_SPLIT=mode,
                 **({'EPOS_HIP_LIB': _ref_lib()} if mode == '0' else {})),
        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
  ckpt = weights.random_init(num_objs=num_objs, seed=3, randomize_bn=True, logits_std=0.2)
  img = np.random.RandomState(0).randint(0, 256, (1, h, w, 3)).astype('f')
  with torch.no_grad(), net_ref.precision(torch.float64):
    ref, ep = net_ref.logits(img, ckpt, num_objs, 64)
  outs = {}
  for mode, (path, pr) in procs.items():
    out, _ = pr.communicate(timeout=600)
    assert pr.returncode == 0, out
    outs[mode] = dict(np.load(path))
  exact = {k: v.permute(0, 2, 3, 1) for k, v in ref.items()}
  # the plan's confidence buffers hold the softmaxed values (model.py:677-678)
  exact['pred_obj_conf'] = torch.softmax(exact['pred_obj_conf'], dim=-1)
  fc = exact['pred_frag_conf']
  exact['pred_frag_conf'] = torch.softmax(
      fc.reshape(fc.shape[:3] + (num_objs, 64)), dim=-1).reshape(fc.shape)
  exact = {k: v.numpy() for k, v in exact.items()}
  exact['decoder'] = ep['decoder/decoder_conv1'].permute(0, 2, 3, 1).numpy()
  rms = {}
  for mode in outs:
    rms[mode] = {k: float(np.sqrt(np.mean((outs[mode][k].reshape(exact[k].shape)
                                           .astype(np.float64) - exact[k]) ** 2)))
                 for k in exact}
  print('rms error vs the fp64 oracle: split', rms['1'], ' fp32 MFMA', rms['0'])
  assert not np.array_equal(outs['1']['decoder'], outs['0']['decoder'])   # two kernels ran
  for k in exact:
    assert rms['1'][k] <= rms['0'][k] * 1.15, k


def test_folded_image_pooling_and_slot_clear_match_the_separate_launches(monkeypatch):
  """Round 4: the image-pooling mean from the last encoder GEMM's 32-row block sums and the slot
  table cleared by the opening im2col launch (the defaults) against the separate kernels
  (EPOS_POOL_FOLD=0, EPOS_AMAX_CLEAR_FOLD=0): the pooled vector agrees to fp32 rounding of a
  4800-term sum, every head to 1e-5 of its scale, the encoder output bit for bit, and the
  plan is two launches shorter."""
  from epos_amd import model, weights
  num_objs, h, w = 2, 96, 128
  ckpt = weights.random_init(num_objs=num_objs, seed=3, randomize_bn=True, logits_std=0.2)
  img = torch.from_numpy(
      np.random.RandomState(0).randint(0, 256, (1, h, w, 3)).astype('f')).cuda()
  mo = model.ModelOptions(model.get_outputs_to_num_channels(num_objs, 64))
  monkeypatch.setenv('EPOS_POOL_FOLD', '0'); monkeypatch.setenv('EPOS_AMAX_CLEAR_FOLD', '0')
  net0 = model.get_net(ckpt, 1, h, w, num_objs, 64, mo, instance=30)
  out0 = {k: v.clone() for k, v in net0.forward(img).items()}
  monkeypatch.setenv('EPOS_POOL_FOLD', '1'); monkeypatch.setenv('EPOS_AMAX_CLEAR_FOLD', '1')
  net1 = model.get_net(ckpt, 1, h, w, num_objs, 64, mo, instance=31)
  assert net1.pool_folded and not net0.pool_folded
  assert len(net1.ops) == len(net0.ops) - 1 and net1.ops[0][0].endswith('/im2col')
  for rep in range(2):
    out1 = net1.forward(img, use_graph=rep > 0)
    torch.cuda.synchronize()
    assert torch.equal(net0.encoder, net1.encoder)
    for k in ('pred_obj_conf', 'pred_frag_conf', 'pred_frag_loc'):
      a, b = out0[k].float(), out1[k].float()
      assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max())), k

