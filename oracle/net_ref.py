"""ORACLE (test infrastructure, not product): torch-CPU fp32 restatement of the
EPOS network forward pass, ``epos_lib/model.py::predict`` with the Xception-65
backbone, layer by layer in the reference's own operation order (conv, then
batch-norm, then activation -- nothing folded or fused).

PARITY UNPINNED beyond the slim known-answer tests: TensorFlow 1.12 +
tf.contrib.slim (README.md:27) are not installed and not installable here, the
reference tree holds no golden activations or checkpoints, so this file is
checked only against (i) the integer KATs of
external/slim/nets/resnet_v1_test.py:58-158 (``conv2d_same`` / ``subsample``) and
(ii) the atrous == dense-then-subsample property of the same file (:197-239).

Weights are a dict keyed by TF variable name holding numpy
arrays in TF layout: conv ``weights`` HWIO, ``depthwise_weights`` [3,3,C,1],
``BatchNorm/{gamma,beta,moving_mean,moving_variance}``, ``biases``.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import numpy as np
import torch
import torch.nn.functional as F

XCEPTION_BN_EPS = 1e-3   # feature.py:300-307 (xception arg scope, epsilon 1e-3)
HEAD_BN_EPS = 1e-5       # model.py:194-199 (ASPP), model.py:307-312 (decoder)

PRED_OBJ_CONF = 'pred_obj_conf'      # common.py:24-27
PRED_OBJ_LABEL = 'pred_obj_label'
PRED_FRAG_CONF = 'pred_frag_conf'
PRED_FRAG_LOC = 'pred_frag_loc'


# Arithmetic type of the restatement: fp32 as the reference (default). fp64 (same fp32
# weights and inputs, every operation carried out in double) is the yardstick the tests
# use to ask which of two fp32 implementations is CLOSER to the exact result.
DTYPE = torch.float32


class precision(object):
  """with net_ref.precision(torch.float64): ..."""

  def __init__(self, dtype):
    self.dtype = dtype

  def __enter__(self):
    global DTYPE
    self.prev, DTYPE = DTYPE, self.dtype

  def __exit__(self, *exc):
    global DTYPE
    DTYPE = self.prev


# torch device of the restatement: None = CPU. 'meta' (shapes only, nothing computed) is
# what the structure trace runs on at the full BASELINE sizes.
DEVICE = None


def _t(a):
  t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DTYPE)
  return t.to(DEVICE) if DEVICE else t


# ----------------------------------------------------------------------------
# Structure trace (tests/test_graph_trace.py): while a `trace()` block is active every
# parametrised layer appends a record and every tensor carries a canonical expression over
# layer outputs -- the grammar of tests/golden/tf_recorder.py, whose fixtures
# (tests/golden/graph_*.json) are what the REFERENCE's own graph-building code produces. The
# arithmetic is not touched: tracing only tags tensors.
# ----------------------------------------------------------------------------
TRACE = None
_EXPR, _ALIVE = {}, []


class trace(object):
  """with net_ref.trace() as tr: net_ref.predict(...); tr.layers / tr.outputs"""

  def __enter__(self):
    global TRACE
    self.layers, self.outputs = [], {}
    TRACE = self
    _EXPR.clear()
    del _ALIVE[:]
    return self

  def __exit__(self, *exc):
    global TRACE
    TRACE = None
    _EXPR.clear()
    del _ALIVE[:]


def _tag(t, expr):
  if TRACE is not None:
    _EXPR[id(t)] = expr
    _ALIVE.append(t)           # ids must stay unique while the trace lives
  return t


def _e(t):
  return _EXPR[id(t)]


def _record(kind, x, y, w, stride, rate, padding, scope):
  if TRACE is None or scope is None:
    return y
  kh, kw = int(w.shape[0]), int(w.shape[1])
  cin = int(x.shape[1])
  cout = cin if kind == 'depthwise_conv2d' else int(w.shape[3])
  TRACE.layers.append({
      'scope': scope, 'op': kind, 'kernel': [kh, kw], 'stride': stride, 'rate': rate,
      'padding': padding, 'cin': cin, 'cout': cout, 'bn_eps': None, 'bias': False,
      'input': _e(x), 'out_hw': [int(y.shape[2]), int(y.shape[3])]})
  return _tag(y, 'L:' + scope)


def relu(x):
  y = F.relu(x)
  if TRACE is not None:
    e = _e(x)
    _tag(y, e if e.startswith('relu(') else 'relu(%s)' % e)
  return y


_relu = relu     # for functions with a parameter called `relu`


def add(a, b):
  y = a + b
  if TRACE is not None:
    _tag(y, 'add(%s)' % ','.join(sorted([_e(a), _e(b)])))
  return y


def concat(ts):
  y = torch.cat(ts, dim=1)
  if TRACE is not None:
    _tag(y, 'concat(%s)' % ','.join(_e(t) for t in ts))
  return y


# ----------------------------------------------------------------------------
# TF/slim primitive semantics, on NCHW torch tensors.
# ----------------------------------------------------------------------------
def fixed_padding(x, kernel_size, rate=1):
  """net_xception.py:74-93 / slim resnet_utils.py:105-111: explicit symmetric-ish
  zero padding of (k_eff - 1) total, beg = total // 2."""
  k_eff = kernel_size + (kernel_size - 1) * (rate - 1)
  pad_total = k_eff - 1
  pad_beg = pad_total // 2
  pad_end = pad_total - pad_beg
  y = F.pad(x, (pad_beg, pad_end, pad_beg, pad_end))
  if TRACE is not None:
    _tag(y, 'pad(%s,%d,%d)' % (_e(x), pad_beg, pad_end))
  return y


def _tf_same_pad(x, k, stride, rate):
  """TensorFlow 'SAME' padding (public TF semantics): out = ceil(in / stride),
  pad_total = max((out - 1) * stride + k_eff - in, 0), pad_beg = pad_total // 2."""
  k_eff = k + (k - 1) * (rate - 1)
  pads = []
  for size in (x.shape[3], x.shape[2]):  # W first for F.pad, then H.
    out = -(-size // stride)
    total = max((out - 1) * stride + k_eff - size, 0)
    pads += [total // 2, total - total // 2]
  return F.pad(x, tuple(pads))


def conv2d_raw(x, w_hwio, stride=1, rate=1, padding='SAME', scope=None):
  """slim.conv2d without normalizer/activation. x NCHW, w HWIO. `scope` (the layer's
  variable scope) is only used by the structure trace."""
  w = _t(w_hwio).permute(3, 2, 0, 1).contiguous()
  k = w.shape[2]
  xp = _tf_same_pad(x, k, stride, rate) if padding == 'SAME' else x
  y = F.conv2d(xp, w, stride=stride, dilation=rate)
  return _record('conv2d', x, y, w_hwio, stride, rate, padding, scope)


def depthwise_raw(x, w_hwc1, stride=1, rate=1, padding='SAME', scope=None):
  """Depthwise part of slim.separable_conv2d(num_outputs=None, depth_multiplier=1).
  Weights [kh, kw, C, 1]."""
  c = x.shape[1]
  w = _t(w_hwc1).permute(2, 3, 0, 1).contiguous()  # [C,1,kh,kw]
  k = w.shape[2]
  xp = _tf_same_pad(x, k, stride, rate) if padding == 'SAME' else x
  y = F.conv2d(xp, w, stride=stride, dilation=rate, groups=c)
  return _record('depthwise_conv2d', x, y, w_hwc1, stride, rate, padding, scope)


def conv2d_same_raw(x, w_hwio, stride, rate=1, scope=None):
  """external/slim/nets/resnet_utils.py:77-122 (conv2d_same) without BN: stride 1
  -> SAME; stride > 1 -> explicit fixed padding then VALID."""
  if stride == 1:
    return conv2d_raw(x, w_hwio, 1, rate, 'SAME', scope)
  k = w_hwio.shape[0]
  return conv2d_raw(fixed_padding(x, k, rate), w_hwio, stride, rate, 'VALID', scope)


def subsample(x, factor):
  """external/slim/nets/resnet_utils.py:59-74: max_pool 1x1 stride factor."""
  if factor == 1:
    return x
  y = x[:, :, ::factor, ::factor]
  if TRACE is not None:
    _tag(y, 'subsample(%s,%d)' % (_e(x), factor))
  return y


def batch_norm(x, wts, scope, eps):
  """slim.batch_norm, inference mode: gamma*(x-mean)/sqrt(var+eps)+beta."""
  g = _t(wts[scope + '/BatchNorm/gamma'])
  b = _t(wts[scope + '/BatchNorm/beta'])
  m = _t(wts[scope + '/BatchNorm/moving_mean'])
  v = _t(wts[scope + '/BatchNorm/moving_variance'])
  y = F.batch_norm(x, m, v, g, b, training=False, eps=eps)
  if TRACE is not None:
    rec = TRACE.layers[-1]
    assert rec['scope'] == scope and _e(x) == 'L:' + scope, (rec['scope'], scope)
    rec['bn_eps'] = eps
    _tag(y, _e(x))
  return y


def resize_bilinear_align_corners(x, size_hw):
  """misc.py:94-107: tf.image.resize_bilinear(align_corners=True)."""
  y = F.interpolate(x, size=tuple(size_hw), mode='bilinear', align_corners=True)
  if TRACE is not None:
    same = (int(x.shape[2]), int(x.shape[3])) == tuple(int(v) for v in size_hw)
    _tag(y, _e(x) if same else 'resize(%s,%dx%d)' % (_e(x), size_hw[0], size_hw[1]))
  return y


def scale_dimension(dim, scale):
  """model.py:100-114."""
  return int((float(dim) - 1.0) * scale + 1.0)


# ----------------------------------------------------------------------------
# Xception-65 (net_xception.py).
# ----------------------------------------------------------------------------
def _conv_bn_relu_same(x, wts, scope, stride, eps):
  """resnet_utils.conv2d_same under the xception arg scope (conv + BN + ReLU),
  net_xception.py:460-463."""
  y = conv2d_same_raw(x, wts[scope + '/weights'], stride, scope=scope)
  return relu(batch_norm(y, wts, scope, eps))


def separable_conv2d_same(x, wts, scope, stride, rate, act, eps):
  """net_xception.py:96-194, non-regularised path (:167-182): depthwise(+BN[+act])
  then 1x1(+BN[+act]); stride on the depthwise only; stride>1 -> fixed_padding+VALID."""
  dw_w = wts[scope + '_depthwise/depthwise_weights']
  if stride == 1:
    y = depthwise_raw(x, dw_w, 1, rate, 'SAME', scope + '_depthwise')
  else:
    y = depthwise_raw(fixed_padding(x, 3, rate), dw_w, stride, rate, 'VALID',
                      scope + '_depthwise')
  y = batch_norm(y, wts, scope + '_depthwise', eps)
  if act:
    y = relu(y)
  y = conv2d_raw(y, wts[scope + '_pointwise/weights'], 1, 1, 'SAME', scope + '_pointwise')
  y = batch_norm(y, wts, scope + '_pointwise', eps)
  if act:
    y = relu(y)
  return y


def xception_module(x, wts, scope, depth_list, skip, act_in_sep, stride, rate,
                    unit_rate_list, eps, end_points):
  """net_xception.py:197-323."""
  residual = x
  for i in range(3):
    if not act_in_sep:
      residual = relu(residual)            # :272-276, ReLU before the sep-conv
    residual = separable_conv2d_same(
        residual, wts, '%s/separable_conv%d' % (scope, i + 1),
        stride=stride if i == 2 else 1, rate=rate * unit_rate_list[i],
        act=act_in_sep, eps=eps)
    end_points['%s/separable_conv%d_pointwise' % (scope, i + 1)] = residual
  if skip == 'conv':
    sc = scope + '/shortcut'
    shortcut = conv2d_raw(x, wts[sc + '/weights'], stride, 1, 'SAME', sc)  # :296-302
    shortcut = batch_norm(shortcut, wts, sc, eps)
    out = add(residual, shortcut)
  elif skip == 'sum':
    out = add(residual, x)
  elif skip == 'none':
    out = residual
  else:
    raise ValueError('Unsupported skip connection type.')
  end_points[scope] = out
  return out


def xception_65_blocks(multi_grid=None):
  """net_xception.py:604-648 block table: (scope, depths, skip, act_in_sep,
  num_units, stride, unit_rate_list)."""
  mg = list(multi_grid) if multi_grid else [1, 1, 1]
  one = [1, 1, 1]
  return [
      ('entry_flow/block1', [128, 128, 128], 'conv', False, 1, 2, one),
      ('entry_flow/block2', [256, 256, 256], 'conv', False, 1, 2, one),
      ('entry_flow/block3', [728, 728, 728], 'conv', False, 1, 2, one),
      ('middle_flow/block1', [728, 728, 728], 'sum', False, 16, 1, one),
      ('exit_flow/block1', [728, 1024, 1024], 'conv', False, 1, 2, one),
      ('exit_flow/block2', [1536, 1536, 2048], 'none', True, 1, 1, mg),
  ]


def xception_65(x, wts, output_stride, multi_grid=None, net='xception_65',
                blocks=None):
  """net_xception.py:396-483 (root) + :326-393 (stack_blocks_dense)."""
  eps = XCEPTION_BN_EPS
  end_points = {}
  if output_stride is not None:
    assert output_stride % 2 == 0
    output_stride //= 2                                   # :455-458
  x = _conv_bn_relu_same(x, wts, net + '/entry_flow/conv1_1', 2, eps)
  end_points[net + '/entry_flow/conv1_1'] = x
  x = _conv_bn_relu_same(x, wts, net + '/entry_flow/conv1_2', 1, eps)
  end_points[net + '/entry_flow/conv1_2'] = x
  current_stride, rate = 1, 1
  for (bscope, depths, skip, act, num_units, stride, url) in (
      blocks or xception_65_blocks(multi_grid)):
    for u in range(num_units):
      # Only the last unit of a block carries the block stride (xception_block
      # replicates the same dict, net_xception.py:516-523: every unit gets it).
      unit_stride = stride
      scope = '%s/%s/unit_%d/xception_module' % (net, bscope, u + 1)
      if output_stride is not None and current_stride > output_stride:
        raise ValueError('The target output_stride cannot be reached.')
      if output_stride is not None and current_stride == output_stride:
        x = xception_module(x, wts, scope, depths, skip, act, 1, rate, url, eps,
                            end_points)                   # :380-382
        rate *= unit_stride
      else:
        x = xception_module(x, wts, scope, depths, skip, act, unit_stride, 1,
                            url, eps, end_points)         # :384-385
        current_stride *= unit_stride
  if output_stride is not None and current_stride != output_stride:
    raise ValueError('The target output_stride cannot be reached.')
  return x, end_points


# ----------------------------------------------------------------------------
# ResNet-v1-101 beta variant (net_resnet_v1_beta.py) -- BASELINE config C5.
# ----------------------------------------------------------------------------
RESNET_BN_EPS = 1e-5     # feature.py:282-287


def max_pool_3x3_s2_same(x):
  """slim.max_pool2d(net, 3, stride=2, padding='SAME') (net_resnet_v1_beta.py:190):
  TF SAME padding, padded cells never win (-inf)."""
  pads = []
  for size in (x.shape[3], x.shape[2]):
    out = -(-size // 2)
    total = max((out - 1) * 2 + 3 - size, 0)
    pads += [total // 2, total - total // 2]
  y = F.max_pool2d(F.pad(x, tuple(pads), value=float('-inf')), 3, stride=2)
  if TRACE is not None:
    _tag(y, 'maxpool(%s,3,2,SAME)' % _e(x))
  return y


def _resnet_conv(x, wts, scope, k, stride, rate, relu, eps=RESNET_BN_EPS):
  """slim.conv2d / resnet_utils.conv2d_same under the resnet arg scope
  (conv + BN [+ ReLU])."""
  w = wts[scope + '/weights']
  if k == 1:
    y = conv2d_raw(x, w, stride, 1, 'SAME', scope)
  else:
    y = conv2d_same_raw(x, w, stride, rate, scope)
  y = batch_norm(y, wts, scope, eps)
  return _relu(y) if relu else y


def bottleneck(x, wts, scope, depth, depth_bottleneck, stride, rate, end_points):
  """net_resnet_v1_beta.py:38-93."""
  depth_in = x.shape[1]
  if depth == depth_in:
    shortcut = subsample(x, stride)                                  # :71-72
  else:
    shortcut = _resnet_conv(x, wts, scope + '/shortcut', 1, stride, 1, False)
  r = _resnet_conv(x, wts, scope + '/conv1', 1, 1, 1, True)           # :80-81
  r = _resnet_conv(r, wts, scope + '/conv2', 3, stride, rate, True)   # :82-83
  r = _resnet_conv(r, wts, scope + '/conv3', 1, 1, 1, False)          # :84-85
  end_points[scope + '/conv3'] = r
  out = relu(add(shortcut, r))                                        # :86
  end_points[scope] = out
  return out


def resnet_v1_101_beta_blocks(multi_grid=None):
  """net_resnet_v1_beta.py:494-505: (scope, [(depth, bottleneck, stride, unit_rate)])."""
  mg = list(multi_grid) if multi_grid else [1, 1, 1]

  def block(scope, base, units, stride):
    return (scope, [(base * 4, base, 1, 1)] * (units - 1) +
            [(base * 4, base, stride, 1)])
  return [block('block1', 64, 3, 2), block('block2', 128, 4, 2),
          block('block3', 256, 23, 2),
          ('block4', [(2048, 512, 1, r) for r in mg])]


def resnet_v1_101_beta(x, wts, output_stride, multi_grid=None,
                       net='resnet_v1_101', blocks=None):
  """net_resnet_v1_beta.py:115-204 + slim stack_blocks_dense
  (external/slim/nets/resnet_utils.py:125-219)."""
  end_points = {}
  if output_stride is not None:
    assert output_stride % 4 == 0
    output_stride //= 4                                               # :185-188
  for i, (cout, stride) in enumerate([(64, 2), (64, 1), (128, 1)], 1):   # :108-110
    x = _resnet_conv(x, wts, '%s/conv1_%d' % (net, i), 3, stride, 1, True)
    end_points['%s/conv1_%d' % (net, i)] = x
  x = max_pool_3x3_s2_same(x)                                         # :190
  end_points[net + '/pool1'] = x
  current_stride, rate = 1, 1
  for bscope, units in (blocks or resnet_v1_101_beta_blocks(multi_grid)):
    for u, (depth, db, stride, unit_rate) in enumerate(units):
      scope = '%s/%s/unit_%d/bottleneck_v1' % (net, bscope, u + 1)
      if output_stride is not None and current_stride == output_stride:
        x = bottleneck(x, wts, scope, depth, db, 1, rate * unit_rate, end_points)
        rate *= stride
      else:
        x = bottleneck(x, wts, scope, depth, db, stride, unit_rate, end_points)
        current_stride *= stride
        if output_stride is not None and current_stride > output_stride:
          raise ValueError('The target output_stride cannot be reached.')
  if output_stride is not None and current_stride != output_stride:
    raise ValueError('The target output_stride cannot be reached.')
  return x, end_points


# ----------------------------------------------------------------------------
# DeepLabv3+ meta-architecture (model.py).
# ----------------------------------------------------------------------------
def split_separable_conv2d(x, wts, scope, rate, eps):
  """model.py:51-97: dw3x3(rate)+BN+ReLU then 1x1+BN+ReLU."""
  y = depthwise_raw(x, wts[scope + '_depthwise/depthwise_weights'], 1, rate,
                    scope=scope + '_depthwise')
  y = relu(batch_norm(y, wts, scope + '_depthwise', eps))
  y = conv2d_raw(y, wts[scope + '_pointwise/weights'], scope=scope + '_pointwise')
  return relu(batch_norm(y, wts, scope + '_pointwise', eps))


def _conv1x1_bn_relu(x, wts, scope, eps):
  return relu(batch_norm(conv2d_raw(x, wts[scope + '/weights'], scope=scope), wts,
                         scope, eps))


def aspp(features, wts, atrous_rates, end_points):
  """model.py:213-265."""
  eps = HEAD_BN_EPS
  h, w = features.shape[2], features.shape[3]
  branches = []
  pooled = features.mean(dim=(2, 3), keepdim=True)                 # :220
  if TRACE is not None:
    _tag(pooled, 'mean(%s)' % _e(features))
  pooled = _conv1x1_bn_relu(pooled, wts, 'image_pooling', eps)     # :223-224
  branches.append(resize_bilinear_align_corners(pooled, (h, w)))   # :225-226
  branches.append(_conv1x1_bn_relu(features, wts, 'aspp0', eps))   # :236-237
  for i, rate in enumerate(atrous_rates, 1):                       # :239-253
    branches.append(split_separable_conv2d(features, wts, 'aspp%d' % i, rate,
                                           eps))
  cat = concat(branches)                                           # :256
  end_points['aspp_concat'] = cat
  out = _conv1x1_bn_relu(cat, wts, 'concat_projection', eps)       # :257-258
  end_points['concat_projection'] = out
  return out                                  # dropout = identity (:259-263)


def decoder(features, low_level, wts, im_size_wh, decoder_output_stride,
            end_points):
  """model.py:268-393 with decoder_use_separable_conv=True (common.py:136-138)."""
  eps = HEAD_BN_EPS
  x = features
  for stage, stride in enumerate(decoder_output_stride):
    suffix = '_%d' % stage if stage else ''
    proj = _conv1x1_bn_relu(low_level, wts,
                            'decoder/feature_projection0' + suffix, eps)  # :349-352
    dw = scale_dimension(im_size_wh[0], 1.0 / stride)              # :355
    dh = scale_dimension(im_size_wh[1], 1.0 / stride)              # :356
    feats = [resize_bilinear_align_corners(t, (dh, dw)) for t in (x, proj)]
    x = concat(feats)                                              # :372
    end_points['decoder_concat' + suffix] = x
    x = split_separable_conv2d(x, wts, 'decoder/decoder_conv0' + suffix, 1, eps)
    end_points['decoder/decoder_conv0' + suffix] = x
    x = split_separable_conv2d(x, wts, 'decoder/decoder_conv1' + suffix, 1, eps)
    end_points['decoder/decoder_conv1' + suffix] = x
  return x


DECODER_TAP = {  # feature.py:50-66
    'xception_65':
        'xception_65/entry_flow/block2/unit_1/xception_module/'
        'separable_conv2_pointwise',
    'resnet_v1_101_beta':
        'resnet_v1_101/block1/unit_2/bottleneck_v1/conv3',
}


def logits(images, wts, num_objs, num_frags, model_variant='xception_65',
           encoder_output_stride=8, decoder_output_stride=(4,),
           atrous_rates=(12, 24, 36), multi_grid=None, crop_size_wh=None):
  """model.py:461-514 (get_logits). images: float [B,H,W,3] in [0,255] (NHWC)."""
  if model_variant not in DECODER_TAP:
    raise ValueError('oracle covers xception_65 and resnet_v1_101_beta.')
  x = torch.as_tensor(np.asarray(images), dtype=torch.float32).to(DTYPE)
  if DEVICE:
    x = x.to(DEVICE)
  x = x.permute(0, 3, 1, 2).contiguous()
  if crop_size_wh is None:
    crop_size_wh = (x.shape[3], x.shape[2])
  _tag(x, 'input')
  x = (2.0 / 255.0) * x - 1.0                                # feature.py:171-174
  _tag(x, 'preprocess(input)')
  if model_variant == 'xception_65':
    feats, end_points = xception_65(x, wts, encoder_output_stride, multi_grid)
  else:
    feats, end_points = resnet_v1_101_beta(x, wts, encoder_output_stride,
                                           multi_grid)
  end_points['encoder'] = feats
  feats = aspp(feats, wts, atrous_rates, end_points)
  feats = decoder(feats, end_points[DECODER_TAP[model_variant]], wts,
                  crop_size_wh, decoder_output_stride, end_points)
  num_channels = {                                            # common.py:189-203
      PRED_OBJ_CONF: num_objs + 1,
      PRED_FRAG_CONF: num_objs * num_frags,
      PRED_FRAG_LOC: num_objs * num_frags * 3,
  }
  out = {}
  for name in sorted(num_channels):                           # model.py:503
    y = conv2d_raw(feats, wts['logits/%s/weights' % name],    # :449-456
                   scope='logits/' + name)
    e = _e(y) if TRACE is not None else None
    y = y + _t(wts['logits/%s/biases' % name]).view(1, -1, 1, 1)
    if TRACE is not None:
      TRACE.layers[-1]['bias'] = True
      _tag(y, e)
    assert y.shape[1] == num_channels[name]
    out[name] = y
  return out, end_points


def predict(images, wts, num_objs, num_frags=64, **kw):
  """model.py:629-687. Returns NHWC numpy arrays with the reference's shapes:
  pred_obj_conf f32[B,h,w,O+1], pred_obj_label i64[B,h,w],
  pred_frag_conf f32[B,h,w,O,F], pred_frag_loc f32[B,h,w,O,F,3]."""
  with torch.no_grad():
    lg, end_points = logits(images, wts, num_objs, num_frags, **kw)
    b, _, h, w = lg[PRED_OBJ_CONF].shape
    obj = lg[PRED_OBJ_CONF].permute(0, 2, 3, 1)
    frag = lg[PRED_FRAG_CONF].permute(0, 2, 3, 1).reshape(
        b, h, w, num_objs, num_frags)                         # model.py:117-147
    loc = lg[PRED_FRAG_LOC].permute(0, 2, 3, 1).reshape(
        b, h, w, num_objs, num_frags, 3)
    obj_conf = torch.softmax(obj, dim=-1)                     # :677
    frag_conf = torch.softmax(frag, dim=-1)                   # :678
    if TRACE is not None:
      eo, ef, el = (_e(lg[k]) for k in (PRED_OBJ_CONF, PRED_FRAG_CONF, PRED_FRAG_LOC))
      TRACE.outputs = {
          PRED_OBJ_CONF: {'expr': 'softmax(%s)' % eo, 'shape': [b, h, w, num_objs + 1]},
          PRED_OBJ_LABEL: {'expr': 'argmax(softmax(%s))' % eo, 'shape': [b, h, w]},
          PRED_FRAG_CONF: {'expr': 'softmax(reshape(%s,%s))' % (ef, [num_objs, num_frags]),
                           'shape': [b, h, w, num_objs, num_frags]},
          PRED_FRAG_LOC: {'expr': 'reshape(%s,%s)' % (el, [num_objs, num_frags, 3]),
                          'shape': [b, h, w, num_objs, num_frags, 3]}}
    if DEVICE == 'meta':
      return None
    return {
        PRED_OBJ_CONF: obj_conf.numpy(),
        PRED_OBJ_LABEL: torch.argmax(obj_conf, dim=3).numpy(),  # :683, int64
        PRED_FRAG_CONF: frag_conf.numpy(),
        PRED_FRAG_LOC: loc.contiguous().numpy(),
        '_logits': {k: v.permute(0, 2, 3, 1).contiguous().numpy()
                    for k, v in lg.items()},
        '_end_points': end_points,
    }
