"""Pins the TF/slim padding semantics of the net oracle to the known-answer tests
held by the reference (external/slim/nets/resnet_v1_test.py:58-158) and checks the
atrous == dense-then-subsample property (:197-239). CPU only."""
import numpy as np
import torch

from oracle import net_ref


def _grid(n):
  return np.add.outer(np.arange(n), np.arange(n)).astype(np.float32)


def _w():
  return _grid(3).reshape(3, 3, 1, 1)


def test_subsample_kat():
  # resnet_v1_test.py:58-70: 3x3 / 4x4 / 5x5 inputs, factor 2 -> x[::2, ::2].
  for n, exp in [(3, [[0, 2], [2, 4]]), (4, [[0, 2], [2, 4]]),
                 (5, [[0, 2, 4], [2, 4, 6], [4, 6, 8]])]:
    x = torch.from_numpy(_grid(n)).view(1, 1, n, n)
    assert net_ref.subsample(x, 2)[0, 0].tolist() == exp


def test_conv2d_same_even_kat():
  # resnet_v1_test.py:72-109.
  x = torch.from_numpy(_grid(4)).view(1, 1, 4, 4)
  y1 = net_ref.conv2d_raw(x, _w(), 1, 1, 'SAME')[0, 0].numpy()
  assert y1.tolist() == [[14, 28, 43, 26], [28, 48, 66, 37], [43, 66, 84, 46],
                         [26, 37, 46, 22]]
  y2 = net_ref.subsample(net_ref.conv2d_raw(x, _w(), 1, 1, 'SAME'), 2)
  y3 = net_ref.conv2d_same_raw(x, _w(), 2)
  assert y3[0, 0].tolist() == [[14, 43], [43, 84]]
  assert torch.equal(y2, y3)
  y4 = net_ref.conv2d_raw(x, _w(), 2, 1, 'SAME')     # plain TF SAME differs
  assert y4[0, 0].tolist() == [[48, 37], [37, 22]]


def test_conv2d_same_odd_kat():
  # resnet_v1_test.py:111-158.
  x = torch.from_numpy(_grid(5)).view(1, 1, 5, 5)
  y3 = net_ref.conv2d_same_raw(x, _w(), 2)
  assert y3[0, 0].tolist() == [[14, 43, 34], [43, 84, 55], [34, 55, 30]]
  y4 = net_ref.conv2d_raw(x, _w(), 2, 1, 'SAME')
  assert torch.equal(y3, y4)


def test_atrous_equals_dense_subsample_property():
  """resnet_v1_test.py:197-239 pattern on one xception module: running it with
  stride converted to atrous and subsampling the output equals the strided one."""
  rng = np.random.RandomState(0)
  c = 8
  wts = {}
  for i in range(1, 4):
    sc = 'm/separable_conv%d' % i
    wts[sc + '_depthwise/depthwise_weights'] = rng.randn(3, 3, c, 1).astype('f')
    wts[sc + '_pointwise/weights'] = (rng.randn(1, 1, c, c) * .3).astype('f')
    for s in ('_depthwise', '_pointwise'):
      wts[sc + s + '/BatchNorm/gamma'] = rng.uniform(.5, 1.5, c).astype('f')
      wts[sc + s + '/BatchNorm/beta'] = rng.randn(c).astype('f') * .1
      wts[sc + s + '/BatchNorm/moving_mean'] = rng.randn(c).astype('f') * .1
      wts[sc + s + '/BatchNorm/moving_variance'] = rng.uniform(.5, 1.5, c).astype('f')
  x = torch.from_numpy(rng.randn(1, c, 9, 11).astype('f'))
  ep = {}
  strided = net_ref.xception_module(x, wts, 'm', [c] * 3, 'sum' if False else
                                    'none', False, 2, 1, [1, 1, 1], 1e-3, ep)
  dense = net_ref.xception_module(x, wts, 'm', [c] * 3, 'none', False, 1, 1,
                                  [1, 1, 1], 1e-3, ep)
  assert torch.allclose(net_ref.subsample(dense, 2), strided, atol=1e-4,
                        rtol=1e-4)


def test_predict_shapes_and_dtypes():
  from epos_amd import weights
  w = weights.random_init(num_objs=2, num_frags=64, seed=1, randomize_bn=True)
  img = np.random.RandomState(0).randint(0, 256, (1, 64, 96, 3)).astype('f')
  out = net_ref.predict(img, w, num_objs=2, num_frags=64)
  assert out['pred_obj_conf'].shape == (1, 16, 24, 3)
  assert out['pred_obj_label'].shape == (1, 16, 24)
  assert out['pred_obj_label'].dtype == np.int64
  assert out['pred_frag_conf'].shape == (1, 16, 24, 2, 64)
  assert out['pred_frag_loc'].shape == (1, 16, 24, 2, 64, 3)
  np.testing.assert_allclose(out['pred_obj_conf'].sum(-1), 1.0, atol=1e-5)
  np.testing.assert_allclose(out['pred_frag_conf'].sum(-1), 1.0, atol=1e-5)
