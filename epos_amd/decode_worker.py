"""Decoder process of epos_amd.frames.Prefetcher:  python -m epos_amd.decode_worker <shm file>

Reads one JSON job per line on stdin -- {"off": byte offset in the shared staging file,
"shape": [h, w, 3], "dtype": "uint8" | "float32", "spec": loader description} -- decodes the
frame (frames.load_pixels) straight into the shared file and answers "ok" or "err <message>"
on stdout, in job order. Imports numpy and PIL only (no torch, no HIP): PIL's decoders hold
the GIL in this stack (measured: four decoder THREADS take as long as one), so the parallelism
has to come from processes, as the reference's comes from tf.data's C++ threads
(datagen.py:680-705)."""
import json
import sys

import numpy as np


def main():
  from epos_amd import frames
  shm = np.memmap(sys.argv[1], dtype=np.uint8, mode='r+')
  out = sys.stdout
  out.write('ready\n')
  out.flush()
  for line in sys.stdin:
    if not line.strip():
      continue
    try:
      job = json.loads(line)
      dt = np.dtype(job['dtype'])
      n = int(np.prod(job['shape'])) * dt.itemsize
      view = shm[job['off']:job['off'] + n].view(dt).reshape(job['shape'])
      px = frames.load_pixels(tuple(job['spec']))
      np.copyto(view, px, casting='same_kind' if px.dtype == dt else 'unsafe')
      out.write('ok\n')
    except Exception as e:                      # reported to the consumer, which raises
      out.write('err %s: %s\n' % (type(e).__name__, str(e).replace('\n', ' ')))
    out.flush()


if __name__ == '__main__':
  main()
