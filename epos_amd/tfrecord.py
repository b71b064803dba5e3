"""TFRecord / ``tf.Example`` input without TensorFlow -- the reader side of the
reference's input pipeline for inference (``datagen.Dataset``:
``_parse_example_proto`` datagen.py:384-422, ``_parse_and_preprocess`` :424-476,
GT filtering :545-575; records written by scripts/create_tfrecord.py:187-210 with
the helpers of epos_lib/tfrecord.py).

PARITY UNPINNED: TensorFlow is not installable here and the tree holds no
``.tfrecord`` file, so this module is checked by round trips through its own
writer only (tests/test_tfrecord.py). The formats are the public ones:
  * TFRecord framing: u64 length | u32 masked-crc32c(length) | data |
    u32 masked-crc32c(data), little endian;
  * ``tf.train.Example`` protobuf: Example{1: Features{1: map<string, Feature>}},
    Feature{1: BytesList, 2: FloatList, 3: Int64List}, each ``repeated value = 1``
    (float / int64 lists packed or unpacked).

Image decoding uses PIL (JPEG / PNG). Frames taller than
``infer_max_height_before_crop`` are shrunk as misc.py:79-93 does, with a numpy
restatement of ``tf.image.resize_area(align_corners=True)`` (box filter over
[y*s, (y+1)*s), s = (in-1)/(out-1), source indices clamped to the image) -- also
unpinned; YCB-V, LM-O and the T-LESS crops used by EPOS need no resize.
"""
import io
import struct

import numpy as np

# ---------------------------------------------------------------- crc32c ----
_CRC_TABLE = None


def _crc32c(data):
  global _CRC_TABLE
  if _CRC_TABLE is None:
    tab = []
    for i in range(256):
      c = i
      for _ in range(8):
        c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
      tab.append(c)
    _CRC_TABLE = tab
  crc = 0xFFFFFFFF
  tab = _CRC_TABLE
  for b in data:
    crc = tab[(crc ^ b) & 0xFF] ^ (crc >> 8)
  return crc ^ 0xFFFFFFFF


def _masked_crc(data):
  crc = _crc32c(data)
  return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------- framing ------
def read_records(path, verify_crc=False):
  """Yields the raw byte string of every record of a .tfrecord file."""
  with open(path, 'rb') as f:
    while True:
      head = f.read(12)
      if not head:
        return
      if len(head) < 12:
        raise IOError('truncated TFRecord header in %s' % path)
      length, len_crc = struct.unpack('<QI', head)
      if verify_crc and _masked_crc(head[:8]) != len_crc:
        raise IOError('corrupt TFRecord length in %s' % path)
      data = f.read(length)
      tail = f.read(4)
      if len(data) < length or len(tail) < 4:
        raise IOError('truncated TFRecord in %s' % path)
      if verify_crc and _masked_crc(data) != struct.unpack('<I', tail)[0]:
        raise IOError('corrupt TFRecord data in %s' % path)
      yield data


def scan_records(path, verify_crc=False):
  """Yields (file offset of the record's data, data) for every record of a .tfrecord file."""
  off = 0
  for data in read_records(path, verify_crc):
    yield off + 12, data
    off += 12 + len(data) + 4


def write_records(path, records):
  with open(path, 'wb') as f:
    for data in records:
      head = struct.pack('<Q', len(data))
      f.write(head + struct.pack('<I', _masked_crc(head)))
      f.write(data + struct.pack('<I', _masked_crc(data)))


# ------------------------------------------------------------ protobuf ------
def _varint(buf, pos):
  result, shift = 0, 0
  while True:
    b = buf[pos]
    pos += 1
    result |= (b & 0x7F) << shift
    if not b & 0x80:
      return result, pos
    shift += 7


def _fields(buf):
  """Yields (field number, wire type, value) of one protobuf message."""
  pos, n = 0, len(buf)
  while pos < n:
    key, pos = _varint(buf, pos)
    num, wt = key >> 3, key & 7
    if wt == 0:
      val, pos = _varint(buf, pos)
    elif wt == 1:
      val = buf[pos:pos + 8]
      pos += 8
    elif wt == 2:
      ln, pos = _varint(buf, pos)
      val = buf[pos:pos + ln]
      pos += ln
    elif wt == 5:
      val = buf[pos:pos + 4]
      pos += 4
    else:
      raise ValueError('unsupported protobuf wire type %d' % wt)
    yield num, wt, val


def _parse_feature(buf):
  for num, wt, val in _fields(buf):
    if num == 1:                                   # BytesList
      return [bytes(v) for n, _, v in _fields(val) if n == 1]
    if num == 2:                                   # FloatList
      out = []
      for n, w, v in _fields(val):
        if n != 1:
          continue
        if w == 2:
          out += list(struct.unpack('<%df' % (len(v) // 4), v))
        else:
          out.append(struct.unpack('<f', v)[0])
      return out
    if num == 3:                                   # Int64List
      out = []
      for n, w, v in _fields(val):
        if n != 1:
          continue
        if w == 2:
          p = 0
          while p < len(v):
            x, p = _varint(v, p)
            out.append(x - (1 << 64) if x >> 63 else x)
        else:
          out.append(v - (1 << 64) if v >> 63 else v)
      return out
  return []


def parse_example(data):
  """Serialized tf.train.Example -> {feature key: list of bytes / float / int}."""
  feats = {}
  for num, _, features in _fields(data):
    if num != 1:
      continue
    for fnum, _, entry in _fields(features):
      if fnum != 1:
        continue
      key, value = None, b''
      for n, _, v in _fields(entry):
        if n == 1:
          key = bytes(v).decode('utf-8')
        elif n == 2:
          value = v
      if key is not None:
        feats[key] = _parse_feature(value)
  return feats


def locate_bytes_feature(data, key):
  """(offset, length) of the FIRST bytes value of feature ``key`` inside the serialized
  Example ``data`` -- positional walk of the same messages parse_example reads (Example{1:
  Features{1: map entry{1: key, 2: Feature{1: BytesList{1: bytes}}}}}), so that a decoder
  thread can read an encoded image straight from the file (epos_amd/frames.py) without
  parsing the record again. None if the key is absent or not a bytes feature."""
  want = key.encode('utf-8') if isinstance(key, str) else key

  def walk(start, end):
    pos = start
    while pos < end:
      tag, pos = _varint(data, pos)
      num, wt = tag >> 3, tag & 7
      if wt == 0:
        _, pos = _varint(data, pos)
      elif wt == 1:
        pos += 8
      elif wt == 5:
        pos += 4
      elif wt == 2:
        ln, pos = _varint(data, pos)
        yield num, pos, pos + ln
        pos += ln
      else:
        raise ValueError('unsupported protobuf wire type %d' % wt)
  for num, fs, fe in walk(0, len(data)):
    if num != 1:
      continue
    for fnum, es, ee in walk(fs, fe):
      if fnum != 1:
        continue
      k, val = None, None
      for n, a, b in walk(es, ee):
        if n == 1:
          k = bytes(data[a:b])
        elif n == 2:
          val = (a, b)
      if k != want or val is None:
        continue
      for n, a, b in walk(*val):
        if n == 1:                                   # BytesList
          for m, c, d in walk(a, b):
            if m == 1:
              return c, d - c
      return None
  return None


def _enc_varint(x):
  x &= (1 << 64) - 1
  out = bytearray()
  while True:
    b = x & 0x7F
    x >>= 7
    if x:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _enc_ld(num, payload):
  return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def encode_example(features):
  """{key: list of bytes | float | int} -> serialized tf.train.Example (what
  tf.train.Example(...).SerializeToString() produces for the same content)."""
  entries = b''
  for key in sorted(features):
    vals = features[key]
    if not isinstance(vals, (list, tuple)):
      vals = [vals]
    if vals and isinstance(vals[0], (bytes, str)):
      body = b''.join(_enc_ld(1, v.encode() if isinstance(v, str) else v)
                      for v in vals)
      feat = _enc_ld(1, body)
    elif vals and isinstance(vals[0], float):
      feat = _enc_ld(2, _enc_ld(1, struct.pack('<%df' % len(vals), *vals)))
    else:
      feat = _enc_ld(3, _enc_ld(1, b''.join(_enc_varint(int(v)) for v in vals)))
    entries += _enc_ld(1, _enc_ld(1, key.encode()) + _enc_ld(2, feat))
  return _enc_ld(1, entries)


# ------------------------------------------------------------- resizing -----
def _area_weights(n_in, n_out):
  """[n_out, n_in] float32 weights of TensorFlow's ResizeArea kernel with
  align_corners=True: output i averages the input interval [i*s, (i+1)*s),
  s = (n_in - 1) / (n_out - 1), partial cells weighted by their overlap, source
  indices beyond the image clamped to the last pixel."""
  scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
  w = np.zeros((n_out, n_in), np.float32)
  inv = np.float32(1.0) / scale if scale > 0 else np.float32(1.0)
  for i in range(n_out):
    lo, hi = np.float32(i) * scale, np.float32(i + 1) * scale
    if scale == 0:
      w[i, 0] = 1.0
      continue
    j = int(np.floor(lo))
    while j < int(np.ceil(hi)):
      if j < lo:
        part = np.float32(j + 1) - lo
      elif j + 1 > hi:
        part = hi - np.float32(j)
      else:
        part = np.float32(1.0)
      w[i, min(max(j, 0), n_in - 1)] += part * inv
      j += 1
  return w


def resize_area(im, out_h, out_w):
  """misc.resize_image_tf's shrinking branch (misc.py:86-89) on an [H,W,C] float
  image."""
  im = np.asarray(im, np.float32)
  if im.shape[0] == out_h and im.shape[1] == out_w:
    return im
  wy = _area_weights(im.shape[0], out_h)
  wx = _area_weights(im.shape[1], out_w)
  h, w, c = im.shape
  rows = (wy @ im.reshape(h, w * c)).reshape(out_h, w, c)          # rows first
  return np.ascontiguousarray(
      np.tensordot(rows, wx, axes=([1], [1])).transpose(0, 2, 1), np.float32)


# ------------------------------------------------------- sample decoding ----
def _scalar(feats, key, default):
  v = feats.get(key)
  return v[0] if v else default


def crop_offsets(max_offset_h, max_offset_w, crop_seed, scene_id, im_id):
  """datagen.py:451-455: offset ~ U{0..max_offset} per axis, independently per
  frame. TensorFlow's op-level random stream cannot be reproduced without
  TensorFlow, so the draw here is a counter-based one keyed by (crop_seed, scene_id,
  im_id): same distribution, reproducible, and independent of the order in which
  ranks read the file. Frames exactly as large as the crop (YCB-V 640x480, the
  headline configuration) get offset 0 in both, as in the reference."""
  if max_offset_h <= 0 and max_offset_w <= 0:
    return 0, 0
  rng = np.random.Generator(np.random.Philox(key=[
      int(crop_seed) & 0xffffffffffffffff,
      ((int(scene_id) & 0xffffffff) << 32) | (int(im_id) & 0xffffffff)]))
  off_h = int(rng.integers(0, max(max_offset_h, 0) + 1))
  off_w = int(rng.integers(0, max(max_offset_w, 0) + 1))
  return off_h, off_w


def sample_meta(feats, crop_size, max_height_before_crop, obj_ids=None,
                min_visib_fract=0.1, crop_seed=0, crop_offset=None, image_size=None):
  """Everything of an inference sample EXCEPT the pixels (datagen.py:424-476,545-575):
  dict(scene_id, im_id, image_path, K f64[3,3], gt_obj_ids, gt_poses, crop_offset (h, w),
  geometry = (h_orig, w_orig, h_new, w_new, crop_h, crop_w)). image_size = (h, w) of the
  encoded image, used when the record has no image/height, image/width features."""
  h_orig = _scalar(feats, 'image/height', None)
  w_orig = _scalar(feats, 'image/width', None)
  if h_orig is None or w_orig is None:
    if image_size is None:
      from PIL import Image
      with Image.open(io.BytesIO(feats['image/encoded'][0])) as im_:
        image_size = (im_.size[1], im_.size[0])          # header only: no decode
    h_orig = image_size[0] if h_orig is None else h_orig
    w_orig = image_size[1] if w_orig is None else w_orig
  h_orig, w_orig = int(h_orig), int(w_orig)
  h_new = min(max_height_before_crop, h_orig)
  scale = np.float32(h_new) / np.float32(h_orig)
  w_new = int(np.float32(w_orig) * scale)
  crop_w, crop_h = crop_size
  if crop_h > h_new or crop_w > w_new:
    raise ValueError('crop %dx%d larger than the frame %dx%d' % (
        crop_w, crop_h, w_new, h_new))
  scene_id = int(_scalar(feats, 'image/scene_id', -1))
  im_id = int(_scalar(feats, 'image/im_id', -1))
  if crop_offset is None:
    crop_offset = crop_offsets(h_new - crop_h, w_new - crop_w, crop_seed, scene_id,
                               im_id)
  off_h, off_w = crop_offset
  if not (0 <= off_h <= h_new - crop_h and 0 <= off_w <= w_new - crop_w):
    raise ValueError('crop offset (%d, %d) outside the frame' % (off_h, off_w))
  # float32 arithmetic as in the TF graph (datagen.py:463-466)
  fx = np.float32(_scalar(feats, 'image/camera/fx', -1.0)) * scale
  fy = np.float32(_scalar(feats, 'image/camera/fy', -1.0)) * scale
  cx = np.float32(_scalar(feats, 'image/camera/cx', -1.0)) * scale - np.float32(off_w)
  cy = np.float32(_scalar(feats, 'image/camera/cy', -1.0)) * scale - np.float32(off_h)
  K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
  ids = [int(x) for x in feats.get('image/object/id', [])]
  vis = list(feats.get('image/object/visibility', [1.0] * len(ids)))
  keep = [i for i, o in enumerate(ids)
          if (obj_ids is None or o in obj_ids) and          # datagen.py:545-553
          (min_visib_fract is None or vis[i] >= min_visib_fract)]   # :555-575
  path = _scalar(feats, 'image/path', b'')
  return {
      'scene_id': scene_id, 'im_id': im_id, 'crop_offset': (off_h, off_w),
      'image_path': path.decode('utf-8') if isinstance(path, bytes) else path,
      'K': K, 'gt_obj_ids': [ids[i] for i in keep],
      'gt_poses': _gt_poses(feats, ids, keep),
      'geometry': (h_orig, w_orig, h_new, w_new, crop_h, crop_w),
  }


def decode_image(encoded, geometry, crop_offset, out=None):
  """The pixels of an inference sample: decode (PIL: JPEG / PNG) -> [shrink to
  max_height_before_crop, misc.py:79-93] -> crop (misc.crop_image). Returns uint8 [crop_h,
  crop_w, 3] when no resize is involved (every value is exactly what
  tf.cast(decode_image(..), tf.float32) holds, datagen.py:435-436; the cast itself happens
  on the device or in decode_sample), float32 after a resize. ``out``: an array of that
  dtype and shape to decode into (a pinned staging buffer)."""
  from PIL import Image
  h_orig, w_orig, h_new, w_new, crop_h, crop_w = geometry
  off_h, off_w = crop_offset
  with Image.open(io.BytesIO(encoded)) as pil:
    im = np.asarray(pil.convert('RGB'))
  if h_new != h_orig:                      # misc.py:79-93 (shrinking: area filter)
    im = resize_area(im.astype(np.float32), h_new, w_new)
  im = im[off_h:off_h + crop_h, off_w:off_w + crop_w]       # misc.crop_image
  if out is not None:
    np.copyto(out, im, casting='same_kind')
    return out
  return np.ascontiguousarray(im)


def decode_sample(feats, crop_size, max_height_before_crop, obj_ids=None,
                  min_visib_fract=0.1, crop_seed=0, crop_offset=None):
  """One parsed Example -> the inference sample of datagen.py:424-476,545-575:
  dict(scene_id, im_id, image_path, image f32[crop_h, crop_w, 3], K f64[3,3],
  gt_obj_ids list, crop_offset (h, w)). crop_size = (width, height) as the reference
  consumes it (datagen.py:448-449). A frame larger than the crop is cropped at a
  random offset (datagen.py:451-455, see crop_offsets; ``crop_offset=(h, w)`` fixes
  it) and the principal point moves with it (datagen.py:465-466)."""
  meta = sample_meta(feats, crop_size, max_height_before_crop, obj_ids, min_visib_fract,
                     crop_seed, crop_offset)
  im = decode_image(feats['image/encoded'][0], meta['geometry'], meta['crop_offset'])
  meta['image'] = np.ascontiguousarray(im, np.float32)
  return meta


def _gt_poses(feats, ids, keep):
  """Ground-truth poses of the kept instances (only used by --vis, infer.py:207-215):
  unit quaternion (w, x, y, z) = pose/q1..q4 as create_tfrecord.py:187-210 writes them
  (transform.quaternion_from_matrix), translation pose/t1..t3 [mm]."""
  q = [feats.get('image/object/pose/q%d' % i, []) for i in range(1, 5)]
  t = [feats.get('image/object/pose/t%d' % i, []) for i in range(1, 4)]
  if any(len(x) != len(ids) for x in q + t):
    return None
  out = []
  for i in keep:
    w, x, y, z = (float(q[j][i]) for j in range(4))
    n = (w * w + x * x + y * y + z * z) ** 0.5 or 1.0
    w, x, y, z = w / n, x / n, y / n, z / n
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    out.append({'obj_id': ids[i], 'R': R,
                't': np.array([[float(t[0][i])], [float(t[1][i])], [float(t[2][i])]])})
  return out


def load_samples(path, crop_size, max_height_before_crop=480, obj_ids=None,
                 min_visib_fract=0.1, verify_crc=False, crop_seed=0):
  """Iterates the inference samples of one .tfrecord file in file order (the
  reference keeps reading sequential at inference, datagen.py:680-683)."""
  for rec in read_records(path, verify_crc):
    yield decode_sample(parse_example(rec), crop_size, max_height_before_crop,
                        obj_ids, min_visib_fract, crop_seed)
