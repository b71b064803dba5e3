"""TFRecord / tf.Example reader (epos_amd/tfrecord.py): framing CRCs against the
published CRC-32C check value, protobuf round trips, sample decoding (K scaling,
visibility / dataset filtering of datagen.py:545-575). PARITY UNPINNED vs
TensorFlow (not installable); formats are the public ones."""
import io
import struct

import numpy as np
import pytest

from epos_amd import tfrecord


def test_crc32c_known_answer():
  # RFC 3720 / Castagnoli check value for "123456789".
  assert tfrecord._crc32c(b'123456789') == 0xE3069283
  assert tfrecord._crc32c(b'') == 0


def _png_bytes(arr):
  from PIL import Image
  buf = io.BytesIO()
  Image.fromarray(arr).save(buf, format='PNG')
  return buf.getvalue()


def _example(scene_id, im_id, img, ids, vis):
  return {
      'image/scene_id': [scene_id], 'image/im_id': [im_id],
      'image/path': [b'scene/rgb/000001.png'], 'image/encoded': [_png_bytes(img)],
      'image/height': [img.shape[0]], 'image/width': [img.shape[1]],
      'image/channels': [3],
      'image/camera/fx': [1066.778], 'image/camera/fy': [1067.487],
      'image/camera/cx': [312.9869], 'image/camera/cy': [241.3109],
      'image/object/id': ids, 'image/object/visibility': [float(v) for v in vis],
      'image/object/pose/t3': [float(900 + i) for i in range(len(ids))],
  }


def test_example_roundtrip_and_negative_int():
  feats = {'a': [1, -2, 3 << 40], 'b': [0.5, -1.25], 'c': [b'xy', b''],
           'd': [7]}
  back = tfrecord.parse_example(tfrecord.encode_example(feats))
  assert back['a'] == [1, -2, 3 << 40] and back['d'] == [7]
  assert back['b'] == [0.5, -1.25] and back['c'] == [b'xy', b'']


def test_unpacked_repeated_fields_are_accepted():
  # FloatList / Int64List written unpacked (one tag per value) must parse too.
  f32 = struct.pack('<f', 2.5)
  float_list = b'\x0d' + f32 + b'\x0d' + struct.pack('<f', -1.0)   # field 1, wt 5
  feature = tfrecord._enc_ld(2, float_list)
  entry = tfrecord._enc_ld(1, b'k') + tfrecord._enc_ld(2, feature)
  ex = tfrecord._enc_ld(1, tfrecord._enc_ld(1, entry))
  assert tfrecord.parse_example(ex) == {'k': [2.5, -1.0]}


def test_tfrecord_file_roundtrip_and_corruption(tmp_path):
  rng = np.random.RandomState(0)
  imgs = [rng.randint(0, 256, (480, 640, 3)).astype(np.uint8) for _ in range(2)]
  exs = [_example(48, i + 1, imgs[i], [2, 5, 99], [0.9, 0.05, 0.8])
         for i in range(2)]
  path = str(tmp_path / 'ycbv_test.tfrecord')
  tfrecord.write_records(path, [tfrecord.encode_example(e) for e in exs])
  samples = list(tfrecord.load_samples(path, (640, 480), 480,
                                       obj_ids=list(range(1, 22)),
                                       verify_crc=True))
  assert len(samples) == 2
  s = samples[1]
  assert (s['scene_id'], s['im_id']) == (48, 2)
  assert s['image'].dtype == np.float32 and s['image'].shape == (480, 640, 3)
  assert np.array_equal(s['image'], imgs[1].astype(np.float32))
  np.testing.assert_allclose(s['K'], [[1066.778, 0, 312.9869],
                                      [0, 1067.487, 241.3109], [0, 0, 1]],
                             rtol=1e-6)
  assert s['gt_obj_ids'] == [2]        # 5: visibility < 0.1, 99: not in the dataset
  raw = bytearray(open(path, 'rb').read())
  raw[40] ^= 0xFF
  open(path, 'wb').write(bytes(raw))
  with pytest.raises(IOError):
    list(tfrecord.read_records(path, verify_crc=True))


def test_oversized_frame_is_resized_before_the_crop(tmp_path):
  img = np.full((540, 720, 3), 77, np.uint8)
  feats = tfrecord.parse_example(tfrecord.encode_example(
      _example(1, 1, img, [1], [1.0])))
  s = tfrecord.decode_sample(feats, (640, 480), 480)        # 540 -> 480 rows, 720 -> 640
  assert s['image'].shape == (480, 640, 3)
  np.testing.assert_allclose(s["image"], 77.0, rtol=2e-4)   # float32 cell bounds, as TF
  s = tfrecord.decode_sample(feats, (720, 540), 540)
  assert s['image'].shape == (540, 720, 3)


def test_resize_area_align_corners():
  """Shrinking branch of misc.resize_image_tf (tf.image.resize_area, align_corners):
  hand-computed box-filter values, constants preserved, identity at equal size."""
  from epos_amd import tfrecord
  im = np.arange(5 * 5, dtype=np.float32).reshape(5, 5, 1)
  out = tfrecord.resize_area(im, 3, 3)             # s = 4 / 2 = 2
  # output row r averages input rows {2r, 2r+1}; the last one reaches past the
  # image and repeats row 4 (clamped): same for columns
  rows = [(im[0] + im[1]) / 2, (im[2] + im[3]) / 2, im[4]]
  exp = np.stack([np.stack([(r[0] + r[1]) / 2, (r[2] + r[3]) / 2, r[4]]) for r in rows])
  np.testing.assert_allclose(out, exp, rtol=1e-6)
  const = np.full((7, 9, 3), 3.25, np.float32)
  np.testing.assert_allclose(tfrecord.resize_area(const, 4, 5), 3.25, rtol=1e-6)
  assert tfrecord.resize_area(im, 5, 5) is not None
  np.testing.assert_array_equal(tfrecord.resize_area(im, 5, 5), im)
  w = tfrecord._area_weights(11, 4)                # fractional scale 10/3
  np.testing.assert_allclose(w.sum(1), 1.0, rtol=1e-6)


def test_oversized_frame_is_shrunk_and_K_scaled(tmp_path):
  from PIL import Image
  import io
  from epos_amd import tfrecord
  rgb = np.random.RandomState(0).randint(0, 256, (96, 128, 3)).astype(np.uint8)
  buf = io.BytesIO(); Image.fromarray(rgb).save(buf, format='PNG')
  feats = {'image/encoded': [buf.getvalue()], 'image/height': [96], 'image/width': [128],
           'image/camera/fx': [200.0], 'image/camera/fy': [210.0],
           'image/camera/cx': [64.0], 'image/camera/cy': [48.0],
           'image/scene_id': [1], 'image/im_id': [2], 'image/path': [b'x.png']}
  path = str(tmp_path / 't.tfrecord')
  tfrecord.write_records(path, [tfrecord.encode_example(feats)])
  s = next(tfrecord.load_samples(path, (64, 48), max_height_before_crop=48))
  assert s['image'].shape == (48, 64, 3)
  np.testing.assert_allclose(s['K'][0, 0], 100.0)   # fx * 48/96
  np.testing.assert_allclose(s['K'][1, 2], 24.0)
  np.testing.assert_allclose(s['image'], tfrecord.resize_area(rgb.astype(np.float32), 48, 64))


def test_crop_offset_and_principal_point_shift():
  """datagen.py:451-468: a frame larger than the crop is cropped at an offset drawn
  per frame, and cx / cy move by it (fx, fy do not)."""
  rgb = np.random.RandomState(1).randint(0, 256, (60, 90, 3)).astype(np.uint8)
  feats = tfrecord.parse_example(tfrecord.encode_example(
      _example(7, 11, rgb, [1], [1.0])))
  full = rgb.astype(np.float32)
  s = tfrecord.decode_sample(feats, (64, 48), 480, crop_offset=(5, 9))
  np.testing.assert_array_equal(s['image'], full[5:53, 9:73])
  assert s['crop_offset'] == (5, 9)
  np.testing.assert_allclose(s['K'][0, 2], np.float32(312.9869) - 9, rtol=1e-7)
  np.testing.assert_allclose(s['K'][1, 2], np.float32(241.3109) - 5, rtol=1e-7)
  np.testing.assert_allclose(s['K'][0, 0], np.float32(1066.778), rtol=1e-7)
  # drawn offsets: inside the legal range, reproducible, keyed by the frame
  seen = set()
  for im_id in range(40):
    oh, ow = tfrecord.crop_offsets(12, 26, 0, 7, im_id)
    assert 0 <= oh <= 12 and 0 <= ow <= 26
    assert (oh, ow) == tfrecord.crop_offsets(12, 26, 0, 7, im_id)
    seen.add((oh, ow))
  assert len(seen) > 20
  assert tfrecord.crop_offsets(0, 0, 3, 1, 2) == (0, 0)        # frame == crop (YCB-V)
  s2 = tfrecord.decode_sample(feats, (64, 48), 480, crop_seed=3)
  oh, ow = s2['crop_offset']
  np.testing.assert_array_equal(s2['image'], full[oh:oh + 48, ow:ow + 64])
  with pytest.raises(ValueError):
    tfrecord.decode_sample(feats, (64, 48), 480, crop_offset=(13, 0))


def _google_example_class():
  """tf.train.Example's schema (tensorflow/core/example/{example,feature}.proto: public
  field numbers) declared at run time for Google's protobuf library -- an independent
  encoder / decoder of the same wire format."""
  from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
  fd = descriptor_pb2.FileDescriptorProto(name='epos_test_example.proto', package='epostest',
                                          syntax='proto3')
  T = descriptor_pb2.FieldDescriptorProto

  def msg(name, fields):
    m = fd.message_type.add(name=name)
    for fname, num, typ, label, tname in fields:
      f = m.field.add(name=fname, number=num, type=typ, label=label)
      if tname:
        f.type_name = tname
    return m
  msg('BytesList', [('value', 1, T.TYPE_BYTES, T.LABEL_REPEATED, None)])
  msg('FloatList', [('value', 1, T.TYPE_FLOAT, T.LABEL_REPEATED, None)])
  msg('Int64List', [('value', 1, T.TYPE_INT64, T.LABEL_REPEATED, None)])
  feat = msg('Feature', [('bytes_list', 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, '.epostest.BytesList'),
                         ('float_list', 2, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, '.epostest.FloatList'),
                         ('int64_list', 3, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, '.epostest.Int64List')])
  feat.oneof_decl.add(name='kind')
  for f in feat.field:
    f.oneof_index = 0
  feats = msg('Features', [('feature', 1, T.TYPE_MESSAGE, T.LABEL_REPEATED,
                            '.epostest.Features.FeatureEntry')])
  entry = feats.nested_type.add(name='FeatureEntry')
  entry.options.map_entry = True
  entry.field.add(name='key', number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
  entry.field.add(name='value', number=2, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL,
                  type_name='.epostest.Feature')
  msg('Example', [('features', 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, '.epostest.Features')])
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  return message_factory.GetMessageClass(pool.FindMessageTypeByName('epostest.Example'))


def test_example_wire_format_against_googles_protobuf_library():
  """The hand-written tf.Example parser / encoder against Google's protobuf runtime on
  the published schema, both directions: what protobuf serialises (packed repeated
  scalars, map entries in its own order, negative int64 as 10-byte varints) parses to
  the same dict, and what this module encodes protobuf parses back. Pins the wire-format
  half of the TFRecord reader to an independent implementation (TensorFlow itself is not
  installable here)."""
  Example = _google_example_class()
  rng = np.random.RandomState(0)
  img = _png_bytes(rng.randint(0, 256, (12, 16, 3)).astype(np.uint8))
  feats = {
      'image/encoded': [img], 'image/path': [b'scene/rgb/000001.png', b''],
      'image/scene_id': [48], 'image/im_id': [1], 'image/height': [480],
      'image/object/id': [2, 5, 21, -1, 3 << 40],
      'image/camera/fx': [1066.778], 'image/object/visibility': [0.9, 0.05, 1.0],
      'image/object/pose/q1': [float(np.float32(x)) for x in rng.standard_normal(7)],
      'empty/int': [],
  }
  ex = Example()
  for k, v in feats.items():
    f = ex.features.feature[k]
    if v and isinstance(v[0], bytes):
      f.bytes_list.value.extend(v)
    elif v and isinstance(v[0], float):
      f.float_list.value.extend(v)
    else:
      f.int64_list.value.extend(v)
  parsed = tfrecord.parse_example(ex.SerializeToString())
  for k, v in feats.items():
    got = parsed.get(k, [])
    if v and isinstance(v[0], float):
      np.testing.assert_array_equal(np.float32(got), np.float32(v), err_msg=k)
    else:
      assert got == v, k
  # the other direction: this module's encoder, Google's parser
  back = Example()
  back.ParseFromString(tfrecord.encode_example({k: v for k, v in feats.items() if v}))
  for k, v in feats.items():
    if not v:
      continue
    f = back.features.feature[k]
    if isinstance(v[0], bytes):
      assert list(f.bytes_list.value) == v, k
    elif isinstance(v[0], float):
      np.testing.assert_array_equal(np.float32(list(f.float_list.value)), np.float32(v))
    else:
      assert list(f.int64_list.value) == v, k
