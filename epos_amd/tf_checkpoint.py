"""TensorFlow checkpoint (TensorBundle, ``model.ckpt-N.index`` + ``.data-*``) reader
without TensorFlow -- what ``tf.train.Saver.restore`` does for the reference at
scripts/infer.py:670-683 (variables restored by ``var.op.name``, misc.py:159-168).

PARITY UNPINNED: no TensorFlow and no checkpoint file are available here; the
module is exercised by round trips through its own minimal writer
(tests/test_tf_checkpoint.py). Formats are the public ones:
  * ``.index``: an SSTable in the LevelDB table format (data blocks of
    prefix-compressed entries + restart array, 5-byte block trailer = compression
    byte + masked CRC-32C, index block, 48-byte footer ending in the magic
    0xdb4775248b80fb57); key "" -> BundleHeaderProto, every other key = variable
    name -> BundleEntryProto {1: dtype, 2: shape, 3: shard_id, 4: offset, 5: size,
    6: crc32c, 7: slices};
  * ``.data-SSSSS-of-NNNNN``: raw little-endian tensor bytes at (offset, size).
Snappy-compressed index blocks (not what TensorFlow's BundleWriter emits) and
partitioned (sliced) variables raise.
"""
import os
import struct

import numpy as np

from epos_amd.tfrecord import _enc_ld, _enc_varint, _fields, _masked_crc, _varint

TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 6: np.int8,
           9: np.int64, 10: np.bool_, 5: np.int16, 17: np.uint16}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


def _block_handle(buf, pos):
  off, pos = _varint(buf, pos)
  size, pos = _varint(buf, pos)
  return off, size, pos


def _read_block(raw, off, size):
  data = raw[off:off + size]
  ctype = raw[off + size]
  if ctype != 0:
    raise NotImplementedError('compressed SSTable block (type %d)' % ctype)
  n_restarts = struct.unpack('<I', data[-4:])[0]
  end = len(data) - 4 - 4 * n_restarts
  pos, key, out = 0, b'', []
  while pos < end:
    shared, pos = _varint(data, pos)
    unshared, pos = _varint(data, pos)
    vlen, pos = _varint(data, pos)
    key = key[:shared] + bytes(data[pos:pos + unshared])
    pos += unshared
    out.append((key, bytes(data[pos:pos + vlen])))
    pos += vlen
  return out


def read_index(index_path):
  """Returns (header dict, {name: entry dict}) of a TensorBundle .index file."""
  raw = open(index_path, 'rb').read()
  if len(raw) < 48 or struct.unpack('<Q', raw[-8:])[0] != TABLE_MAGIC:
    raise ValueError('%s is not a TensorBundle index (bad magic)' % index_path)
  footer = raw[-48:]
  _, _, pos = _block_handle(footer, 0)              # metaindex (unused)
  ioff, isize, _ = _block_handle(footer, pos)
  entries = {}
  header = {}
  for _, handle in _read_block(raw, ioff, isize):
    boff, bsize, _ = _block_handle(handle, 0)
    for key, val in _read_block(raw, boff, bsize):
      if key == b'':
        for num, _, v in _fields(val):
          header[{1: 'num_shards', 2: 'endianness'}.get(num, num)] = v
        continue
      e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0,
           'slices': 0}
      for num, wt, v in _fields(val):
        if num == 1:
          e['dtype'] = v
        elif num == 2:
          for n2, _, dim in _fields(v):
            if n2 == 2:
              size = 0
              for n3, _, x in _fields(dim):
                if n3 == 1:
                  size = x
              e['shape'].append(size)
        elif num == 3:
          e['shard_id'] = v
        elif num == 4:
          e['offset'] = v
        elif num == 5:
          e['size'] = v
        elif num == 7:
          e['slices'] += 1
      entries[key.decode('utf-8')] = e
  return header, entries


def load_checkpoint(prefix, names=None):
  """``prefix`` = path without the .index / .data suffix (e.g.
  ``<model>/train/model.ckpt-2000000``). Returns {variable name: ndarray}; with
  ``names`` only those (missing ones raise KeyError, as Saver.restore would)."""
  header, entries = read_index(prefix + '.index')
  num_shards = int(header.get('num_shards', 1)) or 1
  if header.get('endianness', 0) not in (0,):
    raise NotImplementedError('big-endian checkpoint')
  shards = {}
  out = {}
  for name in (names if names is not None else sorted(entries)):
    e = entries[name]
    if e['slices']:
      raise NotImplementedError('partitioned variable %s' % name)
    if e['dtype'] not in _DTYPES:
      continue                                     # e.g. string tensors
    sid = e['shard_id']
    if sid not in shards:
      shards[sid] = np.memmap('%s.data-%05d-of-%05d' % (prefix, sid, num_shards),
                              dtype=np.uint8, mode='r')
    buf = shards[sid][e['offset']:e['offset'] + e['size']]
    out[name] = np.frombuffer(bytes(buf), dtype=_DTYPES[e['dtype']]).reshape(
        e['shape']).copy()
  return out


def latest_checkpoint(checkpoint_dir):
  """tf.train.latest_checkpoint: the ``checkpoint`` state file if present, else
  the highest-numbered ``*.index`` (infer.py:670-674)."""
  state = os.path.join(checkpoint_dir, 'checkpoint')
  if os.path.exists(state):
    for line in open(state):
      if line.startswith('model_checkpoint_path:'):
        p = line.split(':', 1)[1].strip().strip('"')
        return p if os.path.isabs(p) else os.path.join(checkpoint_dir, p)
  best, best_step = None, -1
  for f in os.listdir(checkpoint_dir) if os.path.isdir(checkpoint_dir) else []:
    if f.endswith('.index'):
      stem = f[:-len('.index')]
      try:
        step = int(stem.rsplit('-', 1)[1])
      except (IndexError, ValueError):
        step = 0
      if step > best_step:
        best, best_step = os.path.join(checkpoint_dir, stem), step
  return best


def to_epos_checkpoint(variables):
  """Keeps the variables the inference graph restores (misc.py:159-168 restores
  every model variable; optimizer slots '/Momentum' and global_step are not
  part of the forward pass)."""
  return {k: np.asarray(v) for k, v in variables.items()
          if not k.endswith('/Momentum') and k != 'global_step'}


# ------------------------------------------------------------ test writer ---
def write_checkpoint(prefix, variables):
  """Minimal TensorBundle writer (one shard, one uncompressed data block, no
  prefix compression) -- for the round-trip tests and for shipping converted
  weights; not a general replacement of tf.train.Saver."""
  data = bytearray()
  items = []
  for name in sorted(variables):
    arr = np.asarray(variables[name], order='C')
    shape = b''.join(_enc_ld(2, _enc_varint((1 << 3) | 0) + _enc_varint(int(d)))
                     for d in arr.shape)
    raw = arr.tobytes()
    entry = (_enc_varint((1 << 3) | 0) + _enc_varint(_DTYPE_IDS[arr.dtype]) +
             _enc_ld(2, shape) +
             _enc_varint((4 << 3) | 0) + _enc_varint(len(data)) +
             _enc_varint((5 << 3) | 0) + _enc_varint(len(raw)))
    items.append((name.encode('utf-8'), entry))
    data += raw
  header = _enc_varint((1 << 3) | 0) + _enc_varint(1)      # num_shards = 1
  items = [(b'', header)] + items

  def block(entries):
    body = bytearray()
    for k, v in entries:                            # shared = 0 for every key
      body += _enc_varint(0) + _enc_varint(len(k)) + _enc_varint(len(v)) + k + v
    body += struct.pack('<I', 0) + struct.pack('<I', 1)   # one restart at 0
    return bytes(body)

  def with_trailer(b):
    return b + b'\x00' + struct.pack('<I', _masked_crc(b + b'\x00'))
  out = bytearray()
  dblock = block(items)
  d_off, d_size = 0, len(dblock)
  out += with_trailer(dblock)
  mblock = block([])
  m_off, m_size = len(out), len(mblock)
  out += with_trailer(mblock)
  handle = _enc_varint(d_off) + _enc_varint(d_size)
  iblock = block([(items[-1][0] + b'\x00', handle)])
  i_off, i_size = len(out), len(iblock)
  out += with_trailer(iblock)
  footer = (_enc_varint(m_off) + _enc_varint(m_size) + _enc_varint(i_off) +
            _enc_varint(i_size))
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
  out += footer
  with open(prefix + '.index', 'wb') as f:
    f.write(bytes(out))
  with open(prefix + '.data-00000-of-00001', 'wb') as f:
    f.write(bytes(data))
