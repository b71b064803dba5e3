"""Weight containers for the EPOS network.

A *checkpoint* here is a plain ``dict`` keyed by the reference's TensorFlow
variable names (scopes from net_xception.py:176,181,295,302,
372-376, model.py:20-22,224,237,242,89,97,258,324,352,376,383,440-456) holding
numpy arrays in TF layout (conv ``weights`` HWIO, ``depthwise_weights``
[3,3,C,1], BatchNorm ``gamma/beta/moving_mean/moving_variance``, ``biases``).
It can be stored as an ``.npz``.

``variable_specs`` enumerates every variable of a model; ``random_init`` draws
them with the reference's initialisers (used for synthetic benchmarks and tests
-- there is no network access for the released checkpoints).
"""
import numpy as np

XCEPTION_BLOCKS = [  # net_xception.py:604-648: (scope, depths, skip, units)
    ('entry_flow/block1', [128, 128, 128], 'conv', 1),
    ('entry_flow/block2', [256, 256, 256], 'conv', 1),
    ('entry_flow/block3', [728, 728, 728], 'conv', 1),
    ('middle_flow/block1', [728, 728, 728], 'sum', 16),
    ('exit_flow/block1', [728, 1024, 1024], 'conv', 1),
    ('exit_flow/block2', [1536, 1536, 2048], 'none', 1),
]

RESNET101_BLOCKS = [  # net_resnet_v1_beta.py:494-505: (scope, base depth, units)
    ('block1', 64, 3), ('block2', 128, 4), ('block3', 256, 23), ('block4', 512, 3),
]

PRED_OBJ_CONF = 'pred_obj_conf'    # common.py:24-27
PRED_OBJ_LABEL = 'pred_obj_label'
PRED_FRAG_CONF = 'pred_frag_conf'
PRED_FRAG_LOC = 'pred_frag_loc'


def outputs_to_num_channels(num_objs, num_frags):
  """common.py:189-203 (frag_cls_agnostic=False)."""
  return {
      PRED_OBJ_CONF: num_objs + 1,
      PRED_FRAG_CONF: num_objs * num_frags,
      PRED_FRAG_LOC: num_objs * num_frags * 3,
  }


def variable_specs(model_variant='xception_65', num_objs=21, num_frags=64,
                   atrous_rates=(12, 24, 36)):
  """Returns a list of (kind, scope, shape-info) for every layer.

  kind: 'conv' (weights HWIO + BN), 'dw' (depthwise_weights + BN),
        'logits' (weights + biases). Each entry carries the initializer family:
        'backbone' (trunc-normal 0.09, net_xception.py:745,782-783),
        'head_dw' (0.33) / 'head_pw' (0.06) (model.py:56-57),
        'xavier' (slim default for plain slim.conv2d in model.py),
        'logits' (trunc-normal 0.01, model.py:437).
  """
  specs = []
  if model_variant == 'xception_65':
    net = 'xception_65'
    specs.append(('conv', net + '/entry_flow/conv1_1', (3, 3, 3, 32), 'backbone'))
    specs.append(('conv', net + '/entry_flow/conv1_2', (3, 3, 32, 64), 'backbone'))
    cin = 64
    for bscope, depths, skip, units in XCEPTION_BLOCKS:
      for u in range(units):
        scope = '%s/%s/unit_%d/xception_module' % (net, bscope, u + 1)
        c = cin
        for i, d in enumerate(depths):
          sc = '%s/separable_conv%d' % (scope, i + 1)
          specs.append(('dw', sc + '_depthwise', (3, 3, c, 1), 'backbone'))
          specs.append(('conv', sc + '_pointwise', (1, 1, c, d), 'backbone'))
          c = d
        if skip == 'conv':
          specs.append(('conv', scope + '/shortcut', (1, 1, cin, depths[-1]),
                        'backbone'))
        cin = depths[-1]
  elif model_variant == 'resnet_v1_101_beta':
    # net_resnet_v1_beta.py:96-112 (root), :38-93 (bottleneck), :494-505 (blocks);
    # variable scope 'resnet_v1_101' (net_resnet_v1_beta.py:452).
    net = 'resnet_v1_101'
    cin = 3
    for i, cout in enumerate([64, 64, 128], 1):
      specs.append(('conv', '%s/conv1_%d' % (net, i), (3, 3, cin, cout), 'he'))
      cin = cout
    for bscope, base, units in RESNET101_BLOCKS:
      for u in range(units):
        scope = '%s/%s/unit_%d/bottleneck_v1' % (net, bscope, u + 1)
        if cin != base * 4:
          specs.append(('conv', scope + '/shortcut', (1, 1, cin, base * 4), 'he'))
        specs.append(('conv', scope + '/conv1', (1, 1, cin, base), 'he'))
        specs.append(('conv', scope + '/conv2', (3, 3, base, base), 'he'))
        specs.append(('conv', scope + '/conv3', (1, 1, base, base * 4), 'he'))
        cin = base * 4
  else:
    raise ValueError('Unsupported model variant: %s' % model_variant)
  specs.append(('conv', 'image_pooling', (1, 1, cin, 256), 'xavier'))
  specs.append(('conv', 'aspp0', (1, 1, cin, 256), 'xavier'))
  for i, _ in enumerate(atrous_rates, 1):
    specs.append(('dw', 'aspp%d_depthwise' % i, (3, 3, cin, 1), 'head_dw'))
    specs.append(('conv', 'aspp%d_pointwise' % i, (1, 1, cin, 256), 'head_pw'))
  specs.append(('conv', 'concat_projection',
                (1, 1, 256 * (2 + len(atrous_rates)), 256), 'xavier'))
  specs.append(('conv', 'decoder/feature_projection0', (1, 1, 256, 48),
                'xavier'))
  specs.append(('dw', 'decoder/decoder_conv0_depthwise', (3, 3, 304, 1),
                'head_dw'))
  specs.append(('conv', 'decoder/decoder_conv0_pointwise', (1, 1, 304, 256),
                'head_pw'))
  specs.append(('dw', 'decoder/decoder_conv1_depthwise', (3, 3, 256, 1),
                'head_dw'))
  specs.append(('conv', 'decoder/decoder_conv1_pointwise', (1, 1, 256, 256),
                'head_pw'))
  for name, ch in sorted(outputs_to_num_channels(num_objs, num_frags).items()):
    specs.append(('logits', 'logits/' + name, (1, 1, 256, ch), 'logits'))
  return specs


def _trunc_normal(rng, shape, std):
  """tf.truncated_normal_initializer: resample outside 2 sigma."""
  x = rng.standard_normal(size=shape)
  bad = np.abs(x) > 2.0
  while bad.any():
    x[bad] = rng.standard_normal(size=int(bad.sum()))
    bad = np.abs(x) > 2.0
  return (x * std).astype(np.float32)


def random_init(model_variant='xception_65', num_objs=21, num_frags=64, seed=0,
                randomize_bn=False, logits_std=0.01, atrous_rates=(12, 24, 36)):
  """Random-init checkpoint with the reference's initialisers.

  randomize_bn=True draws non-trivial BatchNorm statistics (tests use it so that
  BN folding is actually exercised); False gives slim's defaults gamma=1, beta=0,
  mean=0, var=1.
  """
  rng = np.random.RandomState(seed)
  std = {'backbone': 0.09, 'head_dw': 0.33, 'head_pw': 0.06,
         'logits': logits_std}
  w = {}
  for kind, scope, shape, init in variable_specs(
      model_variant, num_objs, num_frags, atrous_rates):
    if init == 'he':      # slim variance_scaling_initializer (resnet_arg_scope)
      fan_in = shape[0] * shape[1] * shape[2]
      arr = _trunc_normal(rng, shape, np.sqrt(2.0 / fan_in) / 0.8796)
    elif init == 'xavier':
      fan_in = shape[0] * shape[1] * shape[2]
      fan_out = shape[0] * shape[1] * shape[3]
      lim = np.sqrt(6.0 / (fan_in + fan_out))
      arr = rng.uniform(-lim, lim, size=shape).astype(np.float32)
    else:
      arr = _trunc_normal(rng, shape, std[init])
    if kind == 'dw':
      w[scope + '/depthwise_weights'] = arr
    else:
      w[scope + '/weights'] = arr
    if kind == 'logits':
      w[scope + '/biases'] = (
          rng.standard_normal(shape[3]).astype(np.float32) * logits_std
          if randomize_bn else np.zeros(shape[3], np.float32))
      continue
    c = shape[2] if kind == 'dw' else shape[3]
    if randomize_bn:
      # The last BN of a ResNet bottleneck gets a small gamma so that 33 stacked
      # residual units keep activations O(1) at random init.
      g_scale = 0.25 if scope.endswith('bottleneck_v1/conv3') else 1.0
      w[scope + '/BatchNorm/gamma'] = (
          rng.uniform(0.5, 1.5, c) * g_scale).astype(np.float32)
      w[scope + '/BatchNorm/beta'] = (
          rng.standard_normal(c) * 0.1).astype(np.float32)
      w[scope + '/BatchNorm/moving_mean'] = (
          rng.standard_normal(c) * 0.1).astype(np.float32)
      w[scope + '/BatchNorm/moving_variance'] = rng.uniform(
          0.5, 1.5, c).astype(np.float32)
    else:
      w[scope + '/BatchNorm/gamma'] = np.ones(c, np.float32)
      w[scope + '/BatchNorm/beta'] = np.zeros(c, np.float32)
      w[scope + '/BatchNorm/moving_mean'] = np.zeros(c, np.float32)
      w[scope + '/BatchNorm/moving_variance'] = np.ones(c, np.float32)
  return w


def heavy_tailed(w, seed=0, tiny_frac=0.01, tiny_scale=1e-10, lognormal_every=4,
                 sigma=3.0):
  """A copy of checkpoint ``w`` whose every conv / logits matrix has the tails of a TRAINED
  network (weight decay drives many weights far below their column's maximum) that the
  reference's initialisers never draw: ``tiny_frac`` of the entries of every matrix are
  scaled by ``tiny_scale``, and every ``lognormal_every``-th output column gets log-normal
  (sigma) magnitudes, renormalised to the column's l2 norm so that activations keep their
  scale. Used by the tests and by ``bench.py --weights heavy-tailed`` to show that such a
  checkpoint stays on the fp16-pair GEMM (until round 5 one such weight moved the whole layer
  to the 20 % slower bf16 x 6 kernel). Depthwise filters and BatchNorm are left alone."""
  rng = np.random.RandomState(seed + 77)
  out = dict(w)
  for key in sorted(w):
    if not key.endswith('/weights'):
      continue
    a = np.asarray(w[key], np.float64)
    shape = a.shape
    m = a.reshape(-1, shape[-1]).copy()
    cols = np.arange(m.shape[1]) % lognormal_every == lognormal_every - 1
    if cols.any() and m.shape[0] > 1:
      norm = np.sqrt((m[:, cols] ** 2).sum(0, keepdims=True))
      h = m[:, cols] * np.exp(sigma * rng.standard_normal((m.shape[0], int(cols.sum()))))
      m[:, cols] = h * (norm / np.maximum(np.sqrt((h ** 2).sum(0, keepdims=True)), 1e-300))
    m[rng.uniform(size=m.shape) < tiny_frac] *= tiny_scale
    out[key] = m.reshape(shape).astype(np.float32)
  return out


def fold_bn(w, scope, eps, kind):
  """Folds inference-mode BatchNorm into a per-output-channel (scale, bias):
  y = scale * conv(x) + bias, scale = gamma / sqrt(var + eps),
  bias = beta - mean * scale. Computed in float64, returned as float32."""
  g = w[scope + '/BatchNorm/gamma'].astype(np.float64)
  b = w[scope + '/BatchNorm/beta'].astype(np.float64)
  m = w[scope + '/BatchNorm/moving_mean'].astype(np.float64)
  v = w[scope + '/BatchNorm/moving_variance'].astype(np.float64)
  scale = g / np.sqrt(v + eps)
  bias = b - m * scale
  return scale.astype(np.float32), bias.astype(np.float32)


def save_npz(path, w):
  np.savez(path, **{k.replace('/', '|'): v for k, v in w.items()})


def load_npz(path):
  with np.load(path) as z:
    return {k.replace('|', '/'): z[k] for k in z.files}
