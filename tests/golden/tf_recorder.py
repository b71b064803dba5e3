"""A RECORDING stand-in for `tensorflow` / `tf.contrib.slim`, used in the build container
only (tests/golden/make_graph_golden.py) to execute the REFERENCE's own graph-building
code -- /root/reference/epos_lib/{model,feature,net_xception,net_resnet_v1_beta}.py and
/root/reference/external/slim/nets/resnet_utils.py, imported from where they lie -- without
TensorFlow, and to write down the graph they build as a canonical layer list.

It is not a TensorFlow emulation: nothing is computed. A "tensor" is a static shape plus a
canonical EXPRESSION over layer outputs; the slim layer functions implement slim's published
argument handling (arg_scope stack, defaults of conv2d / separable_conv2d / batch_norm /
max_pool2d, "biases only without a normalizer", normalizer then activation, outputs
collected under the variable-scope name) and append one record per parametrised layer.

Expression grammar (shared with oracle/net_ref.py's tracer and epos_amd/net.py's plan trace;
tests/test_graph_trace.py compares the three):
  input | preprocess(input) | L:<scope>            L = the layer's output after its
  relu(e) | add(e1,e2) (operands sorted) | concat(e1,...)        normaliser, before its
  pad(e,beg,end) | resize(e,HxW) | mean(e) | maxpool(e,k,s,PAD) | subsample(e,f)  activation
  reshape(e,[..]) | softmax(e) | argmax(e)

Record fields: scope, op (conv2d | depthwise_conv2d), kernel, stride, rate, padding, cin,
cout, bn_eps (None = no normaliser), bias, input (expression), out_hw.
"""
import collections
import contextlib
import functools
import sys
import types
from unittest import mock


class Recorder(object):
  def __init__(self):
    self.layers = []
    self.scope = []            # variable-scope stack (full names)
    self.collections = collections.defaultdict(list)


REC = Recorder()


# ------------------------------------------------------------------ tensors ---
class Shape(list):
  def as_list(self):
    return list(self)

  def with_rank(self, rank):
    assert len(self) == rank
    return self

  @property
  def ndims(self):
    return len(self)


class Tensor(object):
  def __init__(self, shape, expr, dtype='float32'):
    self.shape = Shape(shape)
    self.expr = expr
    self.dtype = dtype
    self.aliases = []

  def get_shape(self):
    return self.shape

  def set_shape(self, shape):
    for i, s in enumerate(shape):
      if s is not None:
        assert self.shape[i] is None or self.shape[i] == s, (self.shape, shape)
        self.shape[i] = s

  def _add(self, other):
    if isinstance(other, Tensor):
      assert list(self.shape) == list(other.shape), (self.shape, other.shape)
      return Tensor(self.shape, 'add(%s)' % ','.join(sorted([self.expr, other.expr])))
    raise TypeError('Tensor + %r' % (other,))
  __add__ = _add
  __radd__ = _add

  def __rmul__(self, c):         # (2.0 / 255.0) * x
    return Tensor(self.shape, 'mul(%s,%.9g)' % (self.expr, float(c)))

  def __sub__(self, c):          # ... - 1.0
    return Tensor(self.shape, 'sub(%s,%.9g)' % (self.expr, float(c)))


def _canon_preprocess(expr):
  """feature.py:171-174, (2.0 / 255.0) * x - 1.0, wherever it occurs in an expression."""
  return expr.replace('sub(mul(input,%.9g),1)' % (2.0 / 255.0), 'preprocess(input)')


def _conv_out(size, k, stride, rate, padding):
  if padding == 'SAME':
    return -(-size // stride)
  k_eff = k + (k - 1) * (rate - 1)
  return (size - k_eff) // stride + 1


# ---------------------------------------------------------------- arg_scope ---
_SCOPE_STACK = [{}]
_DECORATED = set()


def _key(fn):
  return (getattr(fn, '__module__', None), getattr(fn, '__name__', None))


@contextlib.contextmanager
def arg_scope(list_ops_or_scope, **kwargs):
  if isinstance(list_ops_or_scope, dict):
    assert not kwargs
    _SCOPE_STACK.append(dict(list_ops_or_scope))
  else:
    cur = dict(_SCOPE_STACK[-1])
    for op in list_ops_or_scope:
      k = _key(op)
      assert k in _DECORATED, '%s is not decorated with @add_arg_scope' % (k,)
      merged = dict(cur.get(k, {}))
      merged.update(kwargs)
      cur[k] = merged
    _SCOPE_STACK.append(cur)
  try:
    yield _SCOPE_STACK[-1]
  finally:
    _SCOPE_STACK.pop()


def add_arg_scope(fn):
  @functools.wraps(fn)
  def wrapped(*args, **kwargs):
    scoped = _SCOPE_STACK[-1].get(_key(wrapped))
    if scoped:
      merged = dict(scoped)
      merged.update(kwargs)
      kwargs = merged
    return fn(*args, **kwargs)
  _DECORATED.add(_key(wrapped))
  return wrapped


# ----------------------------------------------------------- variable scope ---
class VarScope(object):
  def __init__(self, name):
    self.name = name
    self.original_name_scope = name + '/'


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, values=None, reuse=None, **_):
  if isinstance(name_or_scope, VarScope):
    full = name_or_scope.name
  else:
    name = name_or_scope if name_or_scope is not None else default_name
    assert name is not None
    full = (REC.scope[-1] + '/' + name) if REC.scope else name
  REC.scope.append(full)
  try:
    yield VarScope(full)
  finally:
    REC.scope.pop()


@contextlib.contextmanager
def name_scope(*_a, **_k):
  yield None


def collect_named_outputs(collections_, alias, outputs):
  if collections_:
    outputs.aliases.append(alias)
    names = [collections_] if isinstance(collections_, str) else list(collections_)
    for c in names:
      REC.collections[c].append((alias, outputs))
  return outputs


def convert_collection_to_dict(collection, clear_collection=False):
  out = collections.OrderedDict(REC.collections.get(collection, []))
  if clear_collection:
    REC.collections.pop(collection, None)
  return out


# ------------------------------------------------------------------- layers ---
def relu(x, name=None):
  if x.expr.startswith('relu('):       # ReLU is idempotent: one canonical spelling
    return Tensor(x.shape, x.expr)
  return Tensor(x.shape, 'relu(%s)' % x.expr)


def _two(v):
  return [v, v] if isinstance(v, int) else [int(v[0]), int(v[1])]


@add_arg_scope
def batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001,
               activation_fn=None, param_initializers=None, param_regularizers=None,
               updates_collections='update_ops', is_training=True, reuse=None,
               variables_collections=None, outputs_collections=None, trainable=True,
               batch_weights=None, fused=None, data_format='NHWC', zero_debias_moving_mean=False,
               scope=None, renorm=False, renorm_clipping=None, renorm_decay=0.99,
               adjustment=None):
  assert not is_training, 'the inference graph must not build training-mode batch norm'
  assert center, 'batch norm without beta is not part of the reference graphs'
  assert activation_fn is None
  inputs._bn = {'eps': float(epsilon), 'scale': bool(scale)}
  return inputs


def _finish(kind, inputs, full, kh, kw, stride, rate, padding, cin, cout, normalizer_fn,
            normalizer_params, activation_fn, biases_initializer, outputs_collections):
  b, h, w = inputs.shape[0], inputs.shape[1], inputs.shape[2]
  out = Tensor([b, _conv_out(h, kh, stride, rate, padding),
                _conv_out(w, kw, stride, rate, padding), cout], 'L:' + full)
  bn_eps, bias = None, False
  if normalizer_fn is not None:
    out = normalizer_fn(out, **(normalizer_params or {}))
    bn = getattr(out, '_bn', None)
    assert bn is not None, 'normalizer_fn is not the recorder\'s batch_norm'
    bn_eps = bn['eps']
    assert bn['scale'], 'batch norm without gamma is not part of the reference graphs'
  elif biases_initializer is not None:
    bias = True
  REC.layers.append(collections.OrderedDict([
      ('scope', full), ('op', kind), ('kernel', [kh, kw]), ('stride', stride),
      ('rate', rate), ('padding', padding), ('cin', cin), ('cout', cout),
      ('bn_eps', bn_eps), ('bias', bias), ('input', _canon_preprocess(inputs.expr)),
      ('out_hw', [out.shape[1], out.shape[2]])]))
  if activation_fn is not None:
    assert activation_fn is relu, 'only ReLU activations occur in the reference graphs'
    out = relu(out)
  return collect_named_outputs(outputs_collections, full, out)


@add_arg_scope
def conv2d(inputs, num_outputs, kernel_size, stride=1, padding='SAME', data_format=None,
           rate=1, activation_fn=relu, normalizer_fn=None, normalizer_params=None,
           weights_initializer=None, weights_regularizer=None, biases_initializer='zeros',
           biases_regularizer=None, reuse=None, variables_collections=None,
           outputs_collections=None, trainable=True, scope=None):
  kh, kw = _two(kernel_size)
  s = _two(stride)
  r = _two(rate)
  assert s[0] == s[1] and r[0] == r[1]
  with variable_scope(scope, 'Conv') as sc:
    return _finish('conv2d', inputs, sc.name, kh, kw, s[0], r[0], padding,
                   inputs.shape[3], num_outputs, normalizer_fn, normalizer_params,
                   activation_fn, biases_initializer, outputs_collections)


@add_arg_scope
def separable_conv2d(inputs, num_outputs, kernel_size, depth_multiplier=1, stride=1,
                     padding='SAME', data_format='NHWC', rate=1, activation_fn=relu,
                     normalizer_fn=None, normalizer_params=None, weights_initializer=None,
                     pointwise_initializer=None, weights_regularizer=None,
                     biases_initializer='zeros', biases_regularizer=None, reuse=None,
                     variables_collections=None, outputs_collections=None, trainable=True,
                     scope=None):
  assert num_outputs is None, ('the reference graphs only use the depthwise half '
                               '(split separable convs)')
  assert depth_multiplier == 1
  kh, kw = _two(kernel_size)
  s = _two(stride)
  r = _two(rate)
  with variable_scope(scope, 'SeparableConv2d') as sc:
    return _finish('depthwise_conv2d', inputs, sc.name, kh, kw, s[0], r[0], padding,
                   inputs.shape[3], inputs.shape[3], normalizer_fn, normalizer_params,
                   activation_fn, biases_initializer, outputs_collections)


@add_arg_scope
def max_pool2d(inputs, kernel_size, stride=2, padding='VALID', data_format='NHWC',
               outputs_collections=None, scope=None):
  kh, kw = _two(kernel_size)
  s = _two(stride)[0]
  assert kh == kw
  b, h, w, c = inputs.shape
  if kh == 1:
    out = Tensor([b, _conv_out(h, 1, s, 1, padding), _conv_out(w, 1, s, 1, padding), c],
                 'subsample(%s,%d)' % (inputs.expr, s))
  else:
    out = Tensor([b, _conv_out(h, kh, s, 1, padding), _conv_out(w, kw, s, 1, padding), c],
                 'maxpool(%s,%d,%d,%s)' % (inputs.expr, kh, s, padding))
  return out


@add_arg_scope
def dropout(inputs, keep_prob=0.5, noise_shape=None, is_training=True,
            outputs_collections=None, scope=None, seed=None):
  assert not is_training
  return inputs


def l2_regularizer(scale, scope=None):
  return ('l2', scale)


def last_dimension(shape, min_rank=1):
  return shape[-1]


# ------------------------------------------------------------------ tf ops ---
def pad(tensor, paddings, mode='CONSTANT', name=None, constant_values=0):
  assert paddings[0] == [0, 0] and paddings[3] == [0, 0] and paddings[1] == paddings[2]
  b, h, w, c = tensor.shape
  beg, end = paddings[1]
  return Tensor([b, h + beg + end, w + beg + end, c],
                'pad(%s,%d,%d)' % (tensor.expr, beg, end))


def reduce_mean(x, axis=None, keepdims=None, name=None, **_):
  assert list(axis) == [1, 2] and keepdims
  return Tensor([x.shape[0], 1, 1, x.shape[3]], 'mean(%s)' % x.expr)


def concat(values, axis, name='concat'):
  assert axis == 3
  for v in values:
    assert v.shape[:3] == values[0].shape[:3], [list(t.shape) for t in values]
  return Tensor(values[0].shape[:3] + [sum(v.shape[3] for v in values)],
                'concat(%s)' % ','.join(v.expr for v in values))


def resize_bilinear(images, size, align_corners=False, name=None):
  assert align_corners, 'every resize of the reference is align_corners=True (misc.py:106)'
  h, w = int(size[0]), int(size[1])
  if [images.shape[1], images.shape[2]] == [h, w]:
    # a resize to the same size with align_corners is the identity map on the grid
    return Tensor(images.shape, images.expr)
  return Tensor([images.shape[0], h, w, images.shape[3]],
                'resize(%s,%dx%d)' % (images.expr, h, w))


def cast(x, dtype=None, name=None):
  return x


def shape(x, name=None, out_type=None):
  return list(x.shape)


def add_n(inputs, name=None):
  assert len(inputs) == 1, 'one logits branch per output (model.py:438-458)'
  return inputs[0]


def reshape(tensor, shape_, name=None):
  shape_ = [int(s) for s in shape_]
  n = 1
  for s in tensor.shape:
    n *= s
  if -1 in shape_:
    known = 1
    for s in shape_:
      known *= s if s != -1 else 1
    shape_[shape_.index(-1)] = n // known
  m = 1
  for s in shape_:
    m *= s
  assert m == n, (tensor.shape, shape_)
  return Tensor(shape_, 'reshape(%s,%s)' % (tensor.expr, shape_[3:]))


def softmax(logits, axis=None, name=None, dim=None):
  assert axis in (None, -1)
  return Tensor(logits.shape, 'softmax(%s)' % logits.expr)


def argmax(input, axis=None, name=None, dimension=None, output_type='int64'):
  assert axis == -1
  return Tensor(input.shape[:-1], 'argmax(%s)' % input.expr, 'int64')


def identity(x, name=None):
  return x


# -------------------------------------------------------------------- flags ---
class _Flags(object):
  def __getattr__(self, name):
    raise AttributeError(name)


class _FlagsModule(object):
  def __init__(self):
    self.FLAGS = _Flags()

  def _define(self, name, default, *a, **k):
    object.__setattr__(self.FLAGS, name, default)

  DEFINE_string = DEFINE_integer = DEFINE_float = DEFINE_boolean = DEFINE_bool = _define
  DEFINE_multi_float = DEFINE_multi_integer = _define

  def DEFINE_list(self, name, default, *a, **k):          # absl: 'a,b' -> ['a', 'b']
    if isinstance(default, str):
      default = default.split(',')
    object.__setattr__(self.FLAGS, name, default)

  def DEFINE_enum(self, name, default, values, *a, **k):
    object.__setattr__(self.FLAGS, name, default)


# ------------------------------------------------------------------ modules ---
class _Loose(types.ModuleType):
  """A module whose unknown attributes are inert mocks (import-time references to parts of
  TensorFlow that the inference graph never calls)."""

  def __getattr__(self, name):
    if name.startswith('__'):
      raise AttributeError(name)
    m = mock.MagicMock(name='%s.%s' % (self.__name__, name))
    setattr(self, name, m)
    return m


def install():
  """Puts the recorder into sys.modules as `tensorflow` (+ the sub-modules the reference
  imports) and returns the module. `tensorflow.contrib.slim.nets.resnet_utils` is the
  reference's own external/slim/nets/resnet_utils.py (slim's source, the same file
  tf.contrib ships)."""
  tf = _Loose('tensorflow')
  slim = _Loose('tensorflow.contrib.slim')
  utils = _Loose('tensorflow.contrib.slim.utils')
  utils.collect_named_outputs = collect_named_outputs
  utils.convert_collection_to_dict = convert_collection_to_dict
  utils.last_dimension = last_dimension
  for fn in (conv2d, separable_conv2d, batch_norm, max_pool2d, dropout):
    setattr(slim, fn.__name__, fn)
  slim.arg_scope = arg_scope
  slim.add_arg_scope = add_arg_scope
  slim.l2_regularizer = l2_regularizer
  slim.utils = utils
  contrib = _Loose('tensorflow.contrib')
  contrib.slim = slim
  tf.contrib = contrib
  nn = _Loose('tensorflow.nn')
  nn.relu = relu
  nn.softmax = softmax
  tf.nn = nn
  image = _Loose('tensorflow.image')
  image.resize_bilinear = resize_bilinear
  tf.image = image
  app = _Loose('tensorflow.app')
  app.flags = _FlagsModule()
  tf.app = app
  tf.flags = app.flags
  for fn in (pad, reduce_mean, concat, cast, shape, add_n, reshape, argmax, identity):
    setattr(tf, fn.__name__, fn)
  tf.variable_scope = variable_scope
  tf.name_scope = name_scope
  tf.Tensor = Tensor
  tf.float32 = 'float32'
  tf.AUTO_REUSE = 'AUTO_REUSE'
  sys.modules['tensorflow'] = tf
  sys.modules['tensorflow.contrib'] = contrib
  sys.modules['tensorflow.contrib.slim'] = slim
  py = _Loose('tensorflow.python')
  ops = _Loose('tensorflow.python.ops')
  sys.modules['tensorflow.python'] = py
  sys.modules['tensorflow.python.ops'] = ops
  sys.modules['tensorflow.python.ops.variables'] = _Loose('tensorflow.python.ops.variables')
  sys.modules['cv2'] = mock.MagicMock()
  return tf


def reset():
  REC.layers = []
  REC.scope = []
  REC.collections = collections.defaultdict(list)
  del _SCOPE_STACK[1:]
