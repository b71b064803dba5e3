"""TensorBundle checkpoint reader (epos_amd/tf_checkpoint.py). PARITY UNPINNED vs
TensorFlow (not installable): round trips through the module's own writer, an
SSTable with prefix-compressed keys and several data blocks built by hand, and the
variable set of a full EPOS checkpoint."""
import os
import struct

import numpy as np
import pytest

from epos_amd import tf_checkpoint as tc
from epos_amd import weights
from epos_amd.tfrecord import _enc_varint, _masked_crc


def test_roundtrip_full_epos_variable_set(tmp_path):
  ckpt = weights.random_init(num_objs=2, seed=0, randomize_bn=True)
  ckpt['global_step'] = np.asarray(2000000, np.int64)
  ckpt['xception_65/entry_flow/conv1_1/weights/Momentum'] = np.zeros(
      (3, 3, 3, 32), np.float32)
  prefix = str(tmp_path / 'model.ckpt-2000000')
  tc.write_checkpoint(prefix, ckpt)
  back = tc.load_checkpoint(prefix)
  assert set(back) == set(ckpt)
  for k in ckpt:
    assert back[k].dtype == np.asarray(ckpt[k]).dtype
    assert np.array_equal(back[k], ckpt[k]), k
  epos = tc.to_epos_checkpoint(back)
  assert 'global_step' not in epos
  assert not any(k.endswith('/Momentum') for k in epos)
  assert set(epos) == set(weights.random_init(num_objs=2, seed=0))
  assert tc.latest_checkpoint(str(tmp_path)) == prefix
  with pytest.raises(KeyError):
    tc.load_checkpoint(prefix, names=['does/not/exist'])


def _block(entries, restart_interval=2):
  """LevelDB-style block WITH prefix compression and several restart points."""
  body, restarts, prev = bytearray(), [], b''
  for i, (k, v) in enumerate(entries):
    shared = 0
    if i % restart_interval == 0:
      restarts.append(len(body))
    else:
      while shared < min(len(k), len(prev)) and k[shared] == prev[shared]:
        shared += 1
    body += (_enc_varint(shared) + _enc_varint(len(k) - shared) +
             _enc_varint(len(v)) + k[shared:] + v)
    prev = k
  for r in restarts:
    body += struct.pack('<I', r)
  body += struct.pack('<I', len(restarts))
  return bytes(body)


def test_reads_prefix_compressed_multi_block_index(tmp_path):
  a = np.arange(6, dtype=np.float32).reshape(2, 3)
  b = np.arange(4, dtype=np.int64)
  c = np.ones((1, 1, 2, 2), np.float32) * 7
  data = a.tobytes() + b.tobytes() + c.tobytes()

  def entry(arr, off):
    shape = b''.join(b'\x12' + _enc_varint(len(d)) + d for d in
                     [b'\x08' + _enc_varint(int(s)) for s in arr.shape])
    return (b'\x08' + _enc_varint({np.dtype('f4'): 1, np.dtype('i8'): 9}[arr.dtype]) +
            b'\x12' + _enc_varint(len(shape)) + shape +
            b'\x20' + _enc_varint(off) + b'\x28' + _enc_varint(arr.nbytes))
  e = [(b'', b'\x08\x01'),
       (b'logits/pred_obj_conf/biases', entry(b, a.nbytes)),
       (b'logits/pred_obj_conf/weights', entry(c, a.nbytes + b.nbytes)),
       (b'xception_65/entry_flow/conv1_1/weights', entry(a, 0))]
  out = bytearray()

  def put(blk):
    off = len(out)
    out.extend(blk + b'\x00' + struct.pack('<I', _masked_crc(blk + b'\x00')))
    return off, len(blk)
  h1 = put(_block(e[:3]))
  h2 = put(_block(e[3:]))
  hm = put(_block([]))
  idx = _block([(b'logits/q', _enc_varint(h1[0]) + _enc_varint(h1[1])),
                (b'y', _enc_varint(h2[0]) + _enc_varint(h2[1]))], 1)
  hi = put(idx)
  footer = (_enc_varint(hm[0]) + _enc_varint(hm[1]) + _enc_varint(hi[0]) +
            _enc_varint(hi[1]))
  out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', tc.TABLE_MAGIC))
  prefix = str(tmp_path / 'model.ckpt-5')
  open(prefix + '.index', 'wb').write(bytes(out))
  open(prefix + '.data-00000-of-00001', 'wb').write(data)
  back = tc.load_checkpoint(prefix)
  assert np.array_equal(back['xception_65/entry_flow/conv1_1/weights'], a)
  assert np.array_equal(back['logits/pred_obj_conf/biases'], b)
  assert np.array_equal(back['logits/pred_obj_conf/weights'], c)


def test_bad_magic_raises(tmp_path):
  p = str(tmp_path / 'x')
  open(p + '.index', 'wb').write(b'\x00' * 64)
  with pytest.raises(ValueError):
    tc.read_index(p + '.index')


def _bundle_protos():
  """BundleHeaderProto / BundleEntryProto / TensorShapeProto / VersionDef
  (tensorflow/core/protobuf/tensor_bundle.proto, framework/tensor_shape.proto,
  framework/versions.proto: the public field numbers) declared at run time for Google's
  protobuf library."""
  from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
  fd = descriptor_pb2.FileDescriptorProto(name='epos_test_bundle.proto', package='epostb',
                                          syntax='proto3')
  T = descriptor_pb2.FieldDescriptorProto

  def msg(name, fields, parent=None):
    m = (parent.nested_type if parent is not None else fd.message_type).add(name=name)
    for fname, num, typ, label, tname in fields:
      f = m.field.add(name=fname, number=num, type=typ, label=label)
      if tname:
        f.type_name = tname
    return m
  O, R = T.LABEL_OPTIONAL, T.LABEL_REPEATED
  shape = msg('TensorShapeProto', [('dim', 2, T.TYPE_MESSAGE, R, '.epostb.TensorShapeProto.Dim'),
                                   ('unknown_rank', 3, T.TYPE_BOOL, O, None)])
  msg('Dim', [('size', 1, T.TYPE_INT64, O, None), ('name', 2, T.TYPE_STRING, O, None)], shape)
  msg('VersionDef', [('producer', 1, T.TYPE_INT32, O, None), ('min_consumer', 2, T.TYPE_INT32, O, None)])
  msg('BundleHeaderProto', [('num_shards', 1, T.TYPE_INT32, O, None),
                            ('endianness', 2, T.TYPE_INT32, O, None),
                            ('version', 3, T.TYPE_MESSAGE, O, '.epostb.VersionDef')])
  msg('BundleEntryProto', [('dtype', 1, T.TYPE_INT32, O, None),
                           ('shape', 2, T.TYPE_MESSAGE, O, '.epostb.TensorShapeProto'),
                           ('shard_id', 3, T.TYPE_INT32, O, None),
                           ('offset', 4, T.TYPE_INT64, O, None),
                           ('size', 5, T.TYPE_INT64, O, None),
                           ('crc32c', 6, T.TYPE_FIXED32, O, None)])
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName('epostb.' + n))  # noqa: E731
  return get('BundleHeaderProto'), get('BundleEntryProto')


def test_index_values_serialised_by_googles_protobuf_library(tmp_path):
  """A two-shard TensorBundle whose header / entry VALUES are serialised by Google's
  protobuf runtime on the published schema -- version sub-message, named dimensions, the
  fixed32 crc32c field, shard ids, a scalar (no dims) and a zero-sized tensor included,
  i.e. the fields BundleWriter really emits -- inside a prefix-compressed multi-block
  SSTable: the reader returns every tensor. Pins the protobuf half of the checkpoint
  reader to an independent implementation."""
  Header, Entry = _bundle_protos()
  rng = np.random.RandomState(1)
  tensors = {
      'global_step': (np.asarray(2000000, np.int64), 0),
      'logits/pred_obj_conf/biases': (rng.standard_normal(22).astype(np.float32), 1),
      'xception_65/entry_flow/conv1_1/BatchNorm/gamma': (rng.standard_normal(32).astype(np.float32), 0),
      'xception_65/entry_flow/conv1_1/weights': (rng.standard_normal((3, 3, 3, 32)).astype(np.float32), 1),
      'zz/empty': (np.zeros((0, 4), np.float32), 0),
  }
  shards = [bytearray(), bytearray()]
  h = Header(num_shards=2, endianness=0)
  h.version.producer = 1
  items = [(b'', h.SerializeToString())]
  for name in sorted(tensors):
    arr, sid = tensors[name]
    e = Entry(dtype={np.dtype('f4'): 1, np.dtype('i8'): 9}[arr.dtype], shard_id=sid,
              offset=len(shards[sid]), size=arr.nbytes, crc32c=0x12345678)
    for i, d in enumerate(arr.shape):
      e.shape.dim.add(size=d, name='d%d' % i if i == 0 else '')
    shards[sid] += arr.tobytes()
    items.append((name.encode(), e.SerializeToString()))
  out = bytearray()

  def put(blk):
    off = len(out)
    out.extend(blk + b'\x00' + struct.pack('<I', _masked_crc(blk + b'\x00')))
    return off, len(blk)
  h1 = put(_block(items[:3], 2))
  h2 = put(_block(items[3:], 16))
  hm = put(_block([]))
  hi = put(_block([(items[2][0] + b'\x00', _enc_varint(h1[0]) + _enc_varint(h1[1])),
                   (b'zzz', _enc_varint(h2[0]) + _enc_varint(h2[1]))], 1))
  footer = (_enc_varint(hm[0]) + _enc_varint(hm[1]) + _enc_varint(hi[0]) + _enc_varint(hi[1]))
  out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', tc.TABLE_MAGIC))
  prefix = str(tmp_path / 'model.ckpt-2000000')
  open(prefix + '.index', 'wb').write(bytes(out))
  for sid in (0, 1):
    open('%s.data-%05d-of-00002' % (prefix, sid), 'wb').write(bytes(shards[sid]))
  back = tc.load_checkpoint(prefix)
  assert set(back) == set(tensors)
  for name, (arr, _) in tensors.items():
    assert back[name].dtype == arr.dtype and back[name].shape == arr.shape, name
    assert np.array_equal(back[name], arr), name
  # and the module's own writer produces entries Google's parser reads back
  tc.write_checkpoint(str(tmp_path / 'own'), {'a/b': np.arange(6, dtype=np.float32).reshape(2, 3)})
  hdr, entries = tc.read_index(str(tmp_path / 'own') + '.index')
  assert hdr['num_shards'] == 1 and entries['a/b']['shape'] == [2, 3]
