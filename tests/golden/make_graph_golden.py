#!/usr/bin/env python
"""Writes tests/golden/graph_*.json: the graph that the REFERENCE's own code builds for the
inference network, as a canonical layer list (tests/golden/tf_recorder.py).

Run in the build container only (needs /root/reference):

    python tests/golden/make_graph_golden.py

What runs is /root/reference/epos_lib/model.py::predict (-> multi_scale_logits -> get_logits
-> feature.extract_features -> net_xception.xception_65 / net_resnet_v1_beta.
resnet_v1_101_beta, external/slim/nets/resnet_utils.py) exactly as scripts/infer.py:644-665
calls it, against a recording stand-in for tensorflow / tf.contrib.slim. The fixtures are
data (layer records + output expressions); no reference source travels.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REFERENCE = '/root/reference'

import tf_recorder as R   # noqa: E402


def build(name, model_variant, width, height, num_objs, num_frags, multi_grid=None):
  R.reset()
  common.FLAGS.model_variant = model_variant
  common.FLAGS.multi_grid = multi_grid
  images = R.Tensor([1, height, width, 3], 'input')
  outputs_to_num_channels = common.get_outputs_to_num_channels(num_objs, num_frags)
  model_options = common.ModelOptions(                       # scripts/infer.py:648-652
      outputs_to_num_channels=outputs_to_num_channels,
      crop_size=[width, height],                            # infer_crop_size = [w, h]
      atrous_rates=common.FLAGS.atrous_rates,
      encoder_output_stride=common.FLAGS.encoder_output_stride)
  predictions = model.predict(                               # scripts/infer.py:655-663
      images=images, model_options=model_options,
      upsample_logits=common.FLAGS.upsample_logits,
      image_pyramid=common.FLAGS.image_pyramid, num_objs=num_objs, num_frags=num_frags,
      frag_cls_agnostic=common.FLAGS.frag_cls_agnostic,
      frag_loc_agnostic=common.FLAGS.frag_loc_agnostic)
  doc = {
      'config': {'name': name, 'model_variant': model_variant, 'width': width,
                 'height': height, 'num_objs': num_objs, 'num_frags': num_frags,
                 'multi_grid': multi_grid,
                 'atrous_rates': list(common.FLAGS.atrous_rates),
                 'encoder_output_stride': common.FLAGS.encoder_output_stride,
                 'decoder_output_stride': [int(x) for x in common.FLAGS.decoder_output_stride]},
      'layers': R.REC.layers,
      'outputs': {k: {'expr': v.expr, 'shape': list(v.shape)}
                  for k, v in sorted(predictions.items())},
  }
  path = os.path.join(HERE, 'graph_%s.json' % name)
  with open(path, 'w') as f:
    json.dump(doc, f, indent=0, separators=(',', ':'))
    f.write('\n')
  print('%s: %d layers -> %s' % (name, len(R.REC.layers), path))


if __name__ == '__main__':
  R.install()
  sys.path.insert(0, os.path.join(REFERENCE, 'external', 'slim'))
  sys.path.insert(0, REFERENCE)
  # tf.contrib.slim.nets.resnet_utils = slim's own file, which the reference vendors
  import nets.resnet_utils as slim_resnet_utils          # pylint: disable=import-error
  nets_mod = R._Loose('tensorflow.contrib.slim.nets')
  nets_mod.resnet_utils = slim_resnet_utils
  sys.modules['tensorflow.contrib.slim.nets'] = nets_mod
  sys.modules['tensorflow.contrib.slim.nets.resnet_utils'] = slim_resnet_utils
  sys.modules['tensorflow'].contrib.slim.nets = nets_mod
  from epos_lib import common, model                     # pylint: disable=import-error
  build('c2_xception65_640x480_o21', 'xception_65', 640, 480, 21, 64)
  build('c4_xception65_720x540_o30', 'xception_65', 720, 540, 30, 64)
  build('c5_resnet101beta_640x480_o15', 'resnet_v1_101_beta', 640, 480, 15, 64)
  build('c5_resnet101beta_640x480_o15_mg124', 'resnet_v1_101_beta', 640, 480, 15, 64,
        multi_grid=[1, 2, 4])
  build('c1_xception65_640x480_o1', 'xception_65', 640, 480, 1, 64)
