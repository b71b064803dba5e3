"""Operator-level mirror of ``epos_lib/model.py`` for MI355X.

``predict`` has the reference's name, argument meaning and result keys
(model.py:629-687); the graph-building half of the reference API
(``multi_scale_logits`` for training, model.py:517-626) is out of scope.
The forward pass itself is the static HIP plan of ``epos_amd.net.EposNet``.
"""
import collections

from epos_amd import net as _net
from epos_amd import weights as W

PRED_OBJ_CONF = W.PRED_OBJ_CONF      # common.py:24-27
PRED_OBJ_LABEL = W.PRED_OBJ_LABEL
PRED_FRAG_CONF = W.PRED_FRAG_CONF
PRED_FRAG_LOC = W.PRED_FRAG_LOC


class ModelOptions(collections.namedtuple('ModelOptions', [
    'outputs_to_num_channels', 'crop_size', 'atrous_rates',
    'encoder_output_stride', 'decoder_output_stride', 'model_variant',
    'multi_grid', 'add_image_level_feature', 'aspp_with_batch_norm',
    'aspp_with_separable_conv', 'decoder_use_separable_conv',
    'logits_kernel_size'])):
  """Immutable network configuration (common.py:206-290). Only the values EPOS
  ships as defaults are supported (common.py:96-154, infer.py:586-591)."""
  __slots__ = ()

  def __new__(cls, outputs_to_num_channels, crop_size=None,
              atrous_rates=(12, 24, 36), encoder_output_stride=8,
              decoder_output_stride=(4,), model_variant='xception_65',
              multi_grid=None):
    return super(ModelOptions, cls).__new__(
        cls, outputs_to_num_channels, crop_size, tuple(atrous_rates),
        encoder_output_stride, tuple(decoder_output_stride), model_variant,
        multi_grid, True, True, True, True, 1)


def get_outputs_to_num_channels(num_objs, num_frags):
  """common.py:189-203."""
  return W.outputs_to_num_channels(num_objs, num_frags)


_NETS = {}


def get_net(checkpoint, batch, height, width, num_objs, num_frags,
            model_options=None, device='cuda:0', instance=0):
  """Returns (and caches) the HIP plan for this checkpoint and input shape.
  ``instance`` distinguishes independent plans (own activation buffers) of the
  same network, e.g. the two halves of a double-buffered pipeline."""
  mo = model_options or ModelOptions(
      get_outputs_to_num_channels(num_objs, num_frags))
  key = (id(checkpoint), batch, height, width, num_objs, num_frags,
         mo.model_variant, mo.atrous_rates, mo.encoder_output_stride,
         mo.decoder_output_stride, tuple(mo.multi_grid or ()), str(device),
         instance)
  if key not in _NETS:
    if len(mo.decoder_output_stride) != 1:
      raise ValueError('one decoder stage only (common.py:127-132).')
    _NETS[key] = _net.EposNet(
        checkpoint, batch, height, width, num_objs, num_frags,
        model_variant=mo.model_variant,
        encoder_output_stride=mo.encoder_output_stride,
        decoder_output_stride=mo.decoder_output_stride[0],
        atrous_rates=mo.atrous_rates, multi_grid=mo.multi_grid, device=device)
  return _NETS[key]


def predict(images, model_options, checkpoint, upsample_logits=False,
            image_pyramid=None, num_objs=None, num_frags=None,
            frag_cls_agnostic=False, frag_loc_agnostic=False, device='cuda:0',
            use_graph=False):
  """model.py:629-687. images: float32 [B,H,W,3] in [0,255] (numpy or tensor).

  Returns {pred_obj_conf f32[B,h,w,O+1], pred_obj_label i64[B,h,w],
  pred_frag_conf f32[B,h,w,O,F], pred_frag_loc f32[B,h,w,O,F,3]} as device
  tensors (views of the plan's buffers)."""
  if upsample_logits:
    raise NotImplementedError('upsample_logits=True (default False, '
                              'common.py:152-154) is out of scope.')
  if image_pyramid not in (None, [1.0], (1.0,)):
    raise NotImplementedError('multi-scale inference (default None, '
                              'common.py:96-98) is out of scope.')
  if frag_cls_agnostic or frag_loc_agnostic:
    raise NotImplementedError('class-agnostic fragment heads are out of scope.')
  b, h, w = images.shape[0], images.shape[1], images.shape[2]
  net = get_net(checkpoint, b, h, w, num_objs, num_frags, model_options, device)
  return net.forward(images, use_graph=use_graph)
