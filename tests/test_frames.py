"""CPU tests of the drop-in's input side (epos_amd/frames.py): the scan pass finds the encoded
image of every record where parse_example finds it, decoder threads fill the staging buffers
with exactly the pixels tfrecord.decode_sample returns, batches come in file order with a
bounded number of buffers, and the last batch is padded."""
import io
import os

import numpy as np
import pytest

from epos_amd import frames as eframes
from epos_amd import tfrecord


def _jpeg(arr, fmt='JPEG'):
  from PIL import Image
  buf = io.BytesIO()
  Image.fromarray(arr).save(buf, format=fmt, quality=90)
  return buf.getvalue()


def _write(path, n, h=96, w=128, fmt='JPEG', big=None):
  rng = np.random.RandomState(1)
  recs = []
  for i in range(n):
    hh, ww = big if (big and i % 2) else (h, w)
    img = rng.randint(0, 256, (hh, ww, 3)).astype(np.uint8)
    recs.append(tfrecord.encode_example({
        'image/scene_id': [48 + i % 2], 'image/im_id': [i], 'image/path': [b'x.jpg'],
        'image/encoded': [_jpeg(img, fmt)], 'image/height': [hh], 'image/width': [ww],
        'image/channels': [3],
        'image/camera/fx': [300.0], 'image/camera/fy': [301.0],
        'image/camera/cx': [64.0], 'image/camera/cy': [48.0],
        'image/object/id': [1, 2, 2, 99], 'image/object/visibility': [0.9, 0.05, 0.7, 1.0]}))
  tfrecord.write_records(path, recs)
  return recs


def test_locate_bytes_feature_agrees_with_parse_example(tmp_path):
  recs = _write(str(tmp_path / 'a.tfrecord'), 3)
  for data in recs:
    off, ln = tfrecord.locate_bytes_feature(data, 'image/encoded')
    assert data[off:off + ln] == tfrecord.parse_example(data)['image/encoded'][0]
    off, ln = tfrecord.locate_bytes_feature(data, 'image/path')
    assert data[off:off + ln] == b'x.jpg'
    assert tfrecord.locate_bytes_feature(data, 'image/nope') is None
    assert tfrecord.locate_bytes_feature(data, 'image/height') is None      # not bytes


@pytest.mark.parametrize('procs', [False, True])
@pytest.mark.parametrize('batch', [1, 4])
def test_prefetcher_delivers_what_decode_sample_returns(tmp_path, batch, procs):
  path = str(tmp_path / 'a.tfrecord')
  _write(path, 11)
  ref = list(tfrecord.load_samples(path, (128, 96), 480, [1, 2, 3], None))
  fr = eframes.scan_tfrecords([path], (128, 96), 480, [1, 2, 3])
  assert len(fr) == 11
  for f, r in zip(fr, ref):
    assert (f.scene_id, f.im_id) == (r['scene_id'], r['im_id'])
    assert np.array_equal(f.K, r['K'])
    assert f.targets == {1: 1, 2: 2}          # no visibility filter at inference; 99 unknown
    assert f.dtype == np.uint8
  feed = eframes.Prefetcher(fr, batch, 96, 128, workers=3, ahead=2, inflight=2, pin=False,
                            processes=procs)
  seen, held = [], []
  for i0, chunk, imgs in feed:
    assert imgs.dtype.is_floating_point is False and tuple(imgs.shape) == (batch, 96, 128, 3)
    for b, f in enumerate(chunk):
      j = min(i0 + b, 10)                     # the last batch repeats its last frame
      assert f.im_id == ref[j]['im_id']
      assert np.array_equal(imgs[b].numpy().astype(np.float32), ref[j]['image'])
    seen.append(i0)
    held.append(i0)
    if len(held) > 2:                         # at most `inflight` steps keep their buffers
      feed.release(held.pop(0))
  assert seen == list(range(0, 11, batch))
  assert len(feed._bufs) <= 2 + 2 + 1


def test_prefetcher_resized_frames_come_as_float32(tmp_path):
  """Frames taller than infer_max_height_before_crop are shrunk by the area filter
  (misc.py:79-93): not byte-valued any more, so their batch is staged as float32."""
  path = str(tmp_path / 'b.tfrecord')
  _write(path, 4, fmt='PNG', big=(192, 256))
  ref = list(tfrecord.load_samples(path, (128, 96), 96, None, None))
  fr = eframes.scan_tfrecords([path], (128, 96), 96, None)
  assert [f.dtype for f in fr] == [np.uint8, np.float32, np.uint8, np.float32]
  feed = eframes.Prefetcher(fr, 2, 96, 128, workers=2, ahead=1, inflight=1, pin=False,
                            processes=True)
  for i0, chunk, imgs in feed:
    assert imgs.dtype.is_floating_point
    for b in range(2):
      assert np.array_equal(imgs[b].numpy(), ref[i0 + b]['image'])
      assert np.array_equal(chunk[b].K, ref[i0 + b]['K'])
    feed.release(i0)


def test_prefetcher_raises_a_decoders_error_and_when_starved(tmp_path):
  path = str(tmp_path / 'c.tfrecord')
  _write(path, 6)
  fr = eframes.scan_tfrecords([path], (128, 96), 480, None)
  sp = fr[2]._loader.spec
  fr[2]._loader.spec = sp[:3] + (sp[3] - 40,) + sp[4:]        # truncated JPEG
  for procs in (False, True):
    feed = eframes.Prefetcher(fr, 1, 96, 128, workers=2, ahead=2, inflight=1, pin=False,
                              processes=procs)
    got = []
    with pytest.raises((IOError, OSError)):
      for i0, _, _ in feed:
        got.append(i0)
        feed.release(i0)
    assert got == [0, 1]
    assert not os.path.exists(feed._staging_path)
  fr = eframes.scan_tfrecords([path], (128, 96), 480, None)
  feed = eframes.Prefetcher(fr, 1, 96, 128, workers=2, ahead=1, inflight=1, pin=False,
                            processes=False)
  with pytest.raises(RuntimeError):
    for i0, _, _ in feed:
      pass                                    # never releases
