"""The network STRUCTURE pinned to the reference's own graph-building code (CPU).

tests/golden/graph_*.json hold the layer list that /root/reference/epos_lib/model.py::
predict builds (xception_65 at the C1 / C2 / C4 shapes, resnet_v1_101_beta at C5 with and
without multi_grid) when it is executed against a recording stand-in for tensorflow /
tf.contrib.slim (tests/golden/tf_recorder.py, make_graph_golden.py): per parametrised layer
its scope, kind, kernel, stride, rate, padding, channel counts, BatchNorm epsilon, bias and
-- as a canonical expression over layer outputs -- what it reads (ReLU placement, residual
adds, explicit paddings, concats, resizes, the decoder tap).

Both readings of that code in this repository must reproduce it:
  * oracle/net_ref.py   (the torch restatement: traced on the 'meta' device, full size);
  * epos_amd/net.py     (the HIP plan: dry_run=True, each launch's record is written from the
                         arguments the launch is built from, fusions included).
This pins order / strides / rates / epsilons / activation placement; the primitive
arithmetic stays pinned by the slim KATs (tests/test_oracle_net.py) and the GPU parity tests.
"""
import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'graph_*.json')))


def _load(path):
  with open(path) as f:
    return json.load(f)


def _ckpt(cfg):
  from epos_amd import weights
  return weights.random_init(num_objs=cfg['num_objs'], num_frags=cfg['num_frags'], seed=0,
                             model_variant=cfg['model_variant'])


def _diff(golden_layers, layers):
  ref = {l['scope']: l for l in golden_layers}
  got = {l['scope']: l for l in layers}
  out = ['missing ' + s for s in ref if s not in got]
  out += ['extra ' + s for s in got if s not in ref]
  for s, l in got.items():
    if s in ref and ref[s] != l:
      out.append('%s: %s' % (s, {k: (ref[s][k], l[k]) for k in l if ref[s].get(k) != l[k]}))
  return out


def test_fixtures_cover_the_baseline_configs():
  names = [os.path.basename(p) for p in GOLDEN]
  assert len(names) >= 5, names
  cfgs = [_load(p)['config'] for p in GOLDEN]
  assert {(c['model_variant'], c['width'], c['height'], c['num_objs']) for c in cfgs} >= {
      ('xception_65', 640, 480, 1), ('xception_65', 640, 480, 21),
      ('xception_65', 720, 540, 30), ('resnet_v1_101_beta', 640, 480, 15)}


@pytest.mark.parametrize('path', GOLDEN, ids=[os.path.basename(p)[6:-5] for p in GOLDEN])
def test_oracle_net_builds_the_reference_graph(path):
  from oracle import net_ref
  g = _load(path)
  cfg = g['config']
  img = np.zeros((1, cfg['height'], cfg['width'], 3), np.float32)
  net_ref.DEVICE = 'meta'
  try:
    with net_ref.trace() as tr:
      net_ref.predict(img, _ckpt(cfg), num_objs=cfg['num_objs'], num_frags=cfg['num_frags'],
                      model_variant=cfg['model_variant'], multi_grid=cfg['multi_grid'],
                      atrous_rates=tuple(cfg['atrous_rates']),
                      encoder_output_stride=cfg['encoder_output_stride'],
                      decoder_output_stride=tuple(cfg['decoder_output_stride']))
  finally:
    net_ref.DEVICE = None
  assert tr.layers == g['layers'], _diff(g['layers'], tr.layers)[:5]   # same ORDER too
  assert tr.outputs == g['outputs']


@pytest.mark.parametrize('path', GOLDEN, ids=[os.path.basename(p)[6:-5] for p in GOLDEN])
def test_hip_plan_builds_the_reference_graph(path):
  from epos_amd import net
  g = _load(path)
  cfg = g['config']
  plan = net.EposNet(_ckpt(cfg), 1, cfg['height'], cfg['width'], cfg['num_objs'],
                     cfg['num_frags'], model_variant=cfg['model_variant'],
                     multi_grid=cfg['multi_grid'], atrous_rates=tuple(cfg['atrous_rates']),
                     dry_run=True)
  d = _diff(g['layers'], plan.trace_layers)
  assert not d, d[:5]
  assert len(plan.trace_layers) == len(g['layers'])
  assert plan.trace_outputs == g['outputs']
  # every parametrised layer is a launch of the plan (or fused into one)
  launched = '+'.join(n for n, _ in plan.ops)
  for l in g['layers']:
    assert l['scope'] in launched, l['scope']


def test_a_structural_change_is_caught():
  """The comparison has teeth: a different BatchNorm epsilon, a moved ReLU, a stride on the
  wrong conv or another decoder tap show up as differences."""
  g = _load([p for p in GOLDEN if 'c1_' in p][0])
  import copy
  for mutate in (
      lambda l: l[10].__setitem__('bn_eps', 1e-5),
      lambda l: l[4].__setitem__('input', l[4]['input'].replace('relu(', '(', 1)),
      lambda l: (l[6].__setitem__('stride', 1), l[7].__setitem__('stride', 2)),
      lambda l: [x.__setitem__('input', x['input'].replace('separable_conv2_pointwise',
                                                           'separable_conv3_pointwise'))
                 for x in l if x['scope'] == 'decoder/feature_projection0']):
    layers = copy.deepcopy(g['layers'])
    mutate(layers)
    assert _diff(g['layers'], layers)


def test_recorder_arg_scope_precedence():
  """slim.arg_scope as the recorder implements it: inner scopes override outer ones, call
  arguments override both, re-entering a captured scope restores it, and only decorated
  functions can be scoped."""
  sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
  import tf_recorder as R
  R.reset()
  x = R.Tensor([1, 16, 16, 8], 'input')
  with R.arg_scope([R.conv2d], normalizer_fn=R.batch_norm, stride=1) as outer:
    with R.arg_scope([R.batch_norm], is_training=False, scale=True, epsilon=1e-3):
      with R.arg_scope([R.conv2d], activation_fn=None) as inner:
        R.conv2d(x, 4, 3, scope='a')
        R.conv2d(x, 4, 1, stride=2, activation_fn=R.relu, scope='b')
      with R.arg_scope(inner):
        with R.arg_scope([R.batch_norm], epsilon=1e-5):
          y = R.conv2d(x, 4, 3, rate=2, scope='c')
  a, b, c = R.REC.layers
  assert (a['bn_eps'], a['stride'], a['out_hw']) == (1e-3, 1, [16, 16])
  assert (b['stride'], b['out_hw'], b['kernel']) == (2, [8, 8], [1, 1])
  assert c['bn_eps'] == 1e-5 and c['rate'] == 2 and y.expr == 'L:c'
  with pytest.raises(AssertionError):
    with R.arg_scope([test_recorder_arg_scope_precedence], x=1):
      pass
  R.reset()
