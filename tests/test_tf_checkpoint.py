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
