"""Input side of the drop-in ``infer.py``: frames are described up front (ids, camera,
targets -- cheap) and their pixels are produced on demand by a small pool of decoder threads
into pinned host buffers, a bounded number of batches ahead of the GPU.

The reference streams its input the same way: a ``tf.data`` pipeline that parses and decodes
``num_parallel_calls`` records at a time and prefetches (datagen.py:384-476, 680-705), feeding
``sess.run`` one image per step (scripts/infer.py:712-739). Round 5's infer.py decoded the
WHOLE input into a host list before the first launch and uploaded pageable float32 frames
synchronously; with this module a step's host work is one non-blocking upload of 0.9 MB of
bytes from pinned memory (the float cast happens on the device: epos_u8_to_f32).

Decoders do no Python-level parsing: the scan pass (``scan_tfrecords``) has already located
every encoded image inside its file, so a decoder's work is ``os.pread`` + the PIL decoder +
one copy. PIL's decoders hold the GIL in this stack (four decoder threads take as long as
one), so the decoders are worker PROCESSES (epos_amd/decode_worker.py: numpy + PIL only)
writing into a staging file in /dev/shm; in-process threads remain as a switch.
"""
import collections
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


class Frame(object):
  """One input frame: scene_id, im_id, K f64[3,3], targets {obj_id: instances}, gt_poses (or
  None), and ``load(out=None)`` -> uint8 or float32 [H,W,3] pixels (decoded on demand)."""
  __slots__ = ('scene_id', 'im_id', 'K', 'targets', 'gt_poses', '_loader', 'dtype')

  def __init__(self, scene_id, im_id, K, targets, loader, gt_poses=None, dtype=np.uint8):
    self.scene_id, self.im_id = scene_id, im_id
    self.K = np.asarray(K, np.float64).reshape(3, 3)
    self.targets = targets
    self.gt_poses = gt_poses
    self._loader = loader
    self.dtype = np.dtype(dtype)

  def load(self, out=None):
    return self._loader(out)

  def image_f32(self):
    """float32 pixels as the reference's tensors hold them (--vis, operator path)."""
    return np.ascontiguousarray(self.load(), np.float32)


def scan_tfrecords(paths, crop_size, max_height_before_crop, obj_ids, crop_seed=0,
                   verify_crc=False):
  """One pass over the .tfrecord files: every Example is parsed EXCEPT its pixels (ids,
  camera, annotated objects, crop geometry: tfrecord.sample_meta) and the position of its
  encoded image in the file is noted. Returns a list of Frame in file order (the reference
  reads sequentially at inference, datagen.py:680-683). min_visib_fract is None as in the
  reference's inference Dataset (scripts/infer.py:614): every annotated instance is a target."""
  from epos_amd import tfrecord
  frames = []
  for path in paths:
    for off, data in tfrecord.scan_records(path, verify_crc):
      feats = tfrecord.parse_example(data)
      loc = tfrecord.locate_bytes_feature(data, 'image/encoded')
      if loc is None:
        raise ValueError('record without image/encoded in %s' % path)
      enc = feats['image/encoded'][0]
      assert data[loc[0]:loc[0] + loc[1]] == enc
      meta = tfrecord.sample_meta(feats, crop_size, max_height_before_crop, obj_ids, None,
                                  crop_seed)
      tg = {}
      for o in meta['gt_obj_ids']:             # instance counts, infer.py:462-463
        tg[o] = tg.get(o, 0) + 1
      geo = meta['geometry']
      resized = geo[2] != geo[0]
      frames.append(Frame(
          meta['scene_id'], meta['im_id'], meta['K'], tg,
          _TfrecordLoader(path, off + loc[0], loc[1], geo, meta['crop_offset']),
          meta['gt_poses'], np.float32 if resized else np.uint8))
  return frames


# ------------------------------------------------------------------ loaders ---
# A loader is a plain (kind, arguments) description of where a frame's pixels come from, so
# that it can be sent to a decoder PROCESS as one line of JSON; load_pixels() carries it out.
_FDS = {}
_FD_LOCK = threading.Lock()


def _fd(path):
  fd = _FDS.get(path)
  if fd is None:
    with _FD_LOCK:
      fd = _FDS.get(path)
      if fd is None:
        fd = os.open(path, os.O_RDONLY)
        _FDS[path] = fd
  return fd


def load_pixels(spec, out=None):
  """Carries out a loader description: returns the frame's pixels (uint8 or float32
  [H,W,3]), written into ``out`` when given."""
  kind = spec[0]
  if kind == 'tfrecord':
    # pread the encoded image from its file, decode, [resize], crop (tfrecord.decode_image)
    from epos_amd import tfrecord
    _, path, offset, length, geometry, crop_offset = spec
    enc = os.pread(_fd(path), length, offset)
    if len(enc) != length:
      raise IOError('short read of an encoded image in %s' % path)
    return tfrecord.decode_image(enc, tuple(geometry), tuple(crop_offset), out)
  if kind == 'file':
    _, path, h, w = spec
    if path.endswith('.npy'):
      img = np.load(path)
    else:
      from PIL import Image
      with Image.open(path) as pil:
        img = np.asarray(pil.convert('RGB'))
    img = img[:h, :w]
    if img.shape[:2] != (h, w):
      raise ValueError('frame %s is %s, expected %dx%d (resize/crop of '
                       'datagen.py:424-476 applies to TFRecord input)' % (
                           path, img.shape, w, h))
  elif kind == 'synthetic':
    from epos_amd import synthetic
    _, index, h, w = spec
    img = synthetic.image(index, h, w).astype(np.uint8)
  else:
    raise ValueError('unknown loader %r' % (kind,))
  if out is not None:
    np.copyto(out, img, casting='unsafe')
    return out
  return img


class _Loader(object):
  __slots__ = ('spec',)

  def __init__(self, *spec):
    self.spec = tuple(spec)

  def __call__(self, out=None):
    return load_pixels(self.spec, out)


def _TfrecordLoader(path, offset, length, geometry, crop_offset):
  return _Loader('tfrecord', path, int(offset), int(length), tuple(int(x) for x in geometry),
                 tuple(int(x) for x in crop_offset))


def frames_from_dir(directory, meta, h, w):
  """--frames <dir>: frames.json entries + images (.npy HxWx3 or anything PIL reads)."""
  out = []
  for m in meta:
    path = os.path.join(directory, m['path'])
    dtype = np.uint8
    if path.endswith('.npy'):
      dtype = np.uint8 if np.load(path, mmap_mode='r').dtype == np.uint8 else np.float32
    out.append(Frame(m.get('scene_id', 0), m['im_id'], m['K'],
                     {int(k): int(v) for k, v in m.get('targets', {}).items()},
                     _Loader('file', path, h, w), dtype=dtype))
  return out


def synthetic_frames(indices, h, w, num_objs, objs_per_image=5):
  """--synthetic N: seeded noise frames (integers 0..255: uploaded as bytes)."""
  from epos_amd import synthetic
  return [Frame(0, i, synthetic.YCBV_K.copy(), synthetic.targets(i, num_objs, objs_per_image),
                _Loader('synthetic', int(i), h, w)) for i in indices]


class _ThreadDecoders(object):
  """Decoder pool of in-process threads (EPOS_DECODE_PROCS=0, and the CPU tests): fine for
  loaders that release the GIL (numpy files, synthetic frames); PIL's decoders do not."""

  def __init__(self, workers, staging):
    self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix='epos-decode')
    self._staging = staging

  def submit(self, spec, off, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    view = self._staging[off:off + n].view(dtype).reshape(shape)

    def job():
      px = load_pixels(spec)
      np.copyto(view, px, casting='same_kind' if px.dtype == np.dtype(dtype) else 'unsafe')
    return self._pool.submit(job).result          # a callable that waits / re-raises

  def close(self):
    self._pool.shutdown(wait=False, cancel_futures=True)


class _ProcessDecoders(object):
  """Decoder pool of worker PROCESSES (epos_amd/decode_worker.py), each fed one JSON line per
  frame over its stdin and answering in order on its stdout; pixels land in the shared staging
  file. Frames go to the workers round-robin and are consumed in submission order."""

  def __init__(self, workers, staging_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
    self._procs = [subprocess.Popen(
        [sys.executable, '-m', 'epos_amd.decode_worker', staging_path], stdin=subprocess.PIPE,
        stdout=subprocess.PIPE, env=env, text=True, bufsize=1) for _ in range(workers)]
    self._ready = False
    self._rr = 0

  def _wait_ready(self):
    if not self._ready:
      for p in self._procs:
        line = p.stdout.readline()
        if line.strip() != 'ready':
          raise RuntimeError('decoder process failed to start: %r' % line)
      self._ready = True

  def submit(self, spec, off, shape, dtype):
    import json
    self._wait_ready()
    p = self._procs[self._rr % len(self._procs)]
    self._rr += 1
    p.stdin.write(json.dumps({'off': int(off), 'shape': list(shape),
                              'dtype': np.dtype(dtype).name, 'spec': list(spec)}) + '\n')
    p.stdin.flush()

    def wait():
      line = p.stdout.readline()                  # answers come in job order per worker
      if not line.startswith('ok'):
        raise IOError('frame decoder: %s' % (line.strip() or 'worker died'))
    return wait

  def close(self):
    for p in self._procs:
      try:
        p.stdin.close()
      except Exception:
        pass
    for p in self._procs:
      try:
        p.wait(timeout=5)
      except Exception:
        p.kill()


class Prefetcher(object):
  """Batches of decoded frames in pinned host memory, ``ahead`` batches in front of the
  consumer. Iterating yields (i0, chunk, images) with images a pinned torch tensor [B,H,W,3]
  (uint8, or float32 if some frame of the batch is not byte-valued); the buffer of a batch
  stays valid until ``release(i0)`` -- the caller releases it when the step that uploaded it
  has been collected (the upload is asynchronous). The last batch is padded by repeating its
  last frame, like the reference's fixed batch shape would need.

  Decoders write into a staging file in /dev/shm (shared with the decoder processes); a batch
  is copied from there into its pinned buffer when it is handed out (0.9 MB per frame, a
  memcpy that releases the GIL). ``processes``: None = decoder processes unless
  EPOS_DECODE_PROCS=0; False = in-process threads."""

  def __init__(self, frames, batch, h, w, workers=None, ahead=6, inflight=4, pin=True,
               processes=None):
    import tempfile
    import torch
    self.frames, self.B, self.h, self.w = frames, batch, h, w
    if workers is None:
      workers = int(os.environ.get('EPOS_DECODE_THREADS', 0)) or min(
          8, max(2, (os.cpu_count() or 2) - 2))
    self.workers = max(1, workers)
    self.ahead = max(1, ahead)
    self.starts = list(range(0, len(frames), batch))
    n_buf = self.ahead + inflight + 1
    self._pin = pin and torch.cuda.is_available()
    self._free = collections.deque(range(n_buf))
    self._bufs = {}                       # (buffer index, dtype) -> pinned tensor, lazily
    self._held = {}                       # i0 -> buffer index
    self._slot_bytes = batch * h * w * 3 * 4
    # staging: one slot per buffer, sized for float32 frames
    shm_dir = '/dev/shm' if os.path.isdir('/dev/shm') else None
    fd, self._staging_path = tempfile.mkstemp(prefix='epos_frames_', dir=shm_dir)
    os.ftruncate(fd, n_buf * self._slot_bytes)
    os.close(fd)
    self._staging = np.memmap(self._staging_path, dtype=np.uint8, mode='r+')
    if processes is None:
      processes = os.environ.get('EPOS_DECODE_PROCS', '1') != '0'
    self.processes = bool(processes)
    self._dec = (_ProcessDecoders(self.workers, self._staging_path) if self.processes
                 else _ThreadDecoders(self.workers, self._staging))
    self._pending = collections.deque()   # (i0, chunk, buffer index, dtype, waits, n_real)
    self._next = 0
    self._closed = False

  def _buffer(self, k, dtype):
    import torch
    key = (k, np.dtype(dtype).name)
    t = self._bufs.get(key)
    if t is None:
      t = torch.empty((self.B, self.h, self.w, 3),
                      dtype=torch.uint8 if np.dtype(dtype) == np.uint8 else torch.float32)
      if self._pin:
        t = t.pin_memory()
      self._bufs[key] = t
    return t

  def _submit(self):
    """Queues decode jobs while a staging buffer is free and fewer than ``ahead`` batches wait."""
    while (self._next < len(self.starts) and self._free and
           len(self._pending) < self.ahead):
      i0 = self.starts[self._next]
      self._next += 1
      chunk = list(self.frames[i0:i0 + self.B])
      n_real = len(chunk)
      while len(chunk) < self.B:
        chunk.append(chunk[-1])
      dtype = np.dtype(np.uint8 if all(f.dtype == np.uint8 for f in chunk) else np.float32)
      k = self._free.popleft()
      frame_bytes = self.h * self.w * 3 * dtype.itemsize
      waits = [self._dec.submit(chunk[b]._loader.spec, k * self._slot_bytes + b * frame_bytes,
                                (self.h, self.w, 3), dtype) for b in range(n_real)]
      self._pending.append((i0, chunk, k, dtype, waits, n_real))

  def __iter__(self):
    try:
      self._submit()
      while self._pending:
        i0, chunk, k, dtype, waits, n_real = self._pending.popleft()
        for w_ in waits:
          w_()                               # re-raises a decoder's exception here
        t = self._buffer(k, dtype)
        n = n_real * self.h * self.w * 3 * dtype.itemsize
        src = self._staging[k * self._slot_bytes:k * self._slot_bytes + n].view(dtype)
        np.copyto(t.numpy().reshape(-1)[:src.size], src)
        for b in range(n_real, self.B):      # padding rows
          t[b].copy_(t[n_real - 1])
        self._held[i0] = k
        self._submit()
        yield i0, chunk, t
        if not self._pending:
          self._submit()
          if not self._pending and self._next < len(self.starts):
            raise RuntimeError('Prefetcher: every staging buffer is held (release() the '
                               'batches whose steps have been collected)')
    finally:
      self.close()

  def release(self, i0):
    k = self._held.pop(i0, None)
    if k is not None and not self._closed:
      self._free.append(k)
      self._submit()

  def close(self):
    if self._closed:
      return
    self._closed = True
    self._dec.close()
    del self._staging
    try:
      os.unlink(self._staging_path)
    except OSError:
      pass
