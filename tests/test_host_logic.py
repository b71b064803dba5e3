"""CPU tests of the host-side logic: the C-ABI library loads and exports every
declared symbol, weight specs, BOP I/O, sharding, and the world_size-2 gloo gather."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
  from epos_amd import _lib
  lib = _lib.load()
  header = open(os.path.join(ROOT, 'include', 'epos_hip.h')).read()
  declared = set(re.findall(r'\b(epos_[a-z0-9_]+)\s*\(', header))
  assert declared, 'no declarations parsed'
  assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
  for name in declared:
    assert hasattr(lib, name), name
  assert lib.epos_abi_version() == 4


def test_pack_pointwise_weights_host():
  import ctypes
  from epos_amd import _lib
  lib = _lib.load()
  k, n = 8, 5
  w = np.arange(k * n, dtype=np.float32).reshape(k, n)
  total = lib.epos_pack_pointwise_weights(None, k, n, None)
  assert total == 8 * 128 * 4          # K padded to 32 -> 8 groups, N padded to 128
  dst = np.empty(total, np.float32)
  lib.epos_pack_pointwise_weights(w.ctypes.data_as(ctypes.c_void_p), k, n,
                                  dst.ctypes.data_as(ctypes.c_void_p))
  p = dst.reshape(8, 128, 4)
  for kk in range(k):
    for nn in range(n):
      assert p[kk // 4, nn, kk % 4] == w[kk, nn]
  assert p[2:].sum() == 0 and p[:, n:].sum() == 0


def test_no_cpu_fallback_without_device():
  """The product fails loudly when there is no HIP device (no oracle fallback)."""
  import torch
  if torch.cuda.is_available():
    pytest.skip('a GPU is present')
  from epos_amd import _lib, fitting, model, weights
  xy = np.random.rand(10, 2); xyz = np.random.rand(10, 3)
  with pytest.raises(_lib.EposError):
    fitting.find6DPoses(xy, xyz, np.eye(3))
  with pytest.raises(_lib.EposError):
    model.get_net({}, 1, 64, 64, 1, 64)


def test_product_never_imports_oracle():
  for dirpath, _, files in os.walk(os.path.join(ROOT, 'epos_amd')):
    for f in files:
      if f.endswith(('.py', '.hip', '.h', '.cpp')):
        src = open(os.path.join(dirpath, f)).read()
        assert 'import oracle' not in src and 'from oracle' not in src, f
        assert 'oracle/' not in src or f == '__init__.py', f


def test_variable_specs_param_count():
  from epos_amd import weights
  specs = weights.variable_specs(num_objs=21, num_frags=64)
  n = 0
  for kind, scope, shape, _ in specs:
    n += int(np.prod(shape))
    c = shape[2] if kind == 'dw' else shape[3]
    n += c if kind == 'logits' else 4 * c
  assert abs(n / 1e6 - 42.2) < 0.5, n           # 42.2 M params
  names = [s[1] for s in specs]
  assert names[-3:] == ['logits/pred_frag_conf', 'logits/pred_frag_loc',
                        'logits/pred_obj_conf']   # sorted order, model.py:503
  assert weights.outputs_to_num_channels(21, 64) == {
      'pred_obj_conf': 22, 'pred_frag_conf': 1344, 'pred_frag_loc': 4032}


def test_bop_io_roundtrip(tmp_path):
  from epos_amd import bop_io
  poses = [{'scene_id': 48, 'im_id': 1, 'obj_id': 5, 'score': 12.5,
            'R': np.arange(9.).reshape(3, 3), 't': np.array([[1.], [2.], [3.]]),
            'time': 0.25}]
  p = str(tmp_path / 'est.csv')
  bop_io.save_bop_results(p, poses)
  lines = open(p).read().split('\n')
  assert lines[0] == 'scene_id,im_id,obj_id,score,R,t,time'
  assert lines[1].startswith('48,1,5,12.5,0.0 1.0 2.0')
  back = bop_io.load_bop_results(p)
  assert np.array_equal(back[0]['R'], poses[0]['R']) and back[0]['time'] == 0.25


def test_shard_range_covers_everything():
  from epos_amd import dist as ed
  for n in [0, 1, 7, 32, 33]:
    for world in [1, 2, 3, 8]:
      seen = []
      for r in range(world):
        b, e = ed.shard_range(n, r, world)
        seen += list(range(b, e))
      assert seen == list(range(n))


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from epos_amd import dist as ed
rank, world, _ = ed.init_from_env(backend='gloo')
b, e = ed.shard_range(5, rank, world)
poses = [{'scene_id': 1, 'im_id': i, 'obj_id': 3 + rank, 'score': float(i),
          'R': np.eye(3) * (i + 1), 't': np.full((3, 1), rank + 0.5),
          'time': 0.1} for i in range(b, e)]
merged = ed.gather_poses(poses, max_records=8)
tmax = ed.max_over_ranks(1.0 + rank)
# uneven shards (N %% world != 0, a rank with nothing): the ranks agree on the record
# count themselves (max_records=None) instead of deriving it from their local shard
uneven = ed.gather_poses(poses if rank == 0 else [], max_records=None)
uneven2 = ed.gather_poses(poses, max_records=None)
ed.barrier()
assert [p['im_id'] for p in uneven] == [0, 1, 2], uneven
assert [p['im_id'] for p in uneven2] == [0, 1, 2, 3, 4], uneven2
if rank == 0:
  assert [p['im_id'] for p in merged] == [0, 1, 2, 3, 4], merged
  assert [p['obj_id'] for p in merged] == [3, 3, 3, 4, 4]
  assert merged[4]['t'][0, 0] == 1.5 and merged[2]['R'][1, 1] == 3.0
  assert tmax == 2.0
  print('GATHER_OK')
dist.destroy_process_group()
'''


def test_two_rank_gloo_gather(tmp_path):
  script = tmp_path / 'worker.py'
  script.write_text(WORKER % {'root': ROOT})
  env = dict(os.environ, MASTER_ADDR='127.0.0.1')
  out = subprocess.run(
      [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
       '--nproc-per-node=2', '--master-addr', '127.0.0.1', '--master-port',
       '29617', str(script)], env=env, capture_output=True, text=True,
      timeout=300)
  assert 'GATHER_OK' in out.stdout, out.stdout + out.stderr
