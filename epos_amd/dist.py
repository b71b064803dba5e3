"""Multi-GPU data parallelism for EPOS inference: images are independent units
(the reference processes them one at a time, infer.py:712-727), so a batch shards
across ranks with NO data-path collective; the only exchange is one gather of
fixed-size pose records at the end of a batch (RCCL over xGMI on the GPU box,
``backend='nccl'``; gloo in the CPU tests).

Record layout (float64, 17 columns):
  scene_id, im_id, obj_id, score, R (9, row major), t (3), time
"""
import os

import numpy as np
import torch
import torch.distributed as dist

RECORD_COLS = 17


def init_from_env(backend=None, force=False):
  """Initialises torch.distributed from the torchrun environment (RANK,
  WORLD_SIZE, MASTER_ADDR, MASTER_PORT). Returns (rank, world, local_rank). A
  single-rank job needs no process group and gets none unless ``force`` (the RCCL
  smoke test initialises a world of one on the one-GPU box)."""
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if (world > 1 or force) and not dist.is_initialized():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
      # EPOS_DIST_BACKEND=gloo: exercise the multi-rank flow where RCCL cannot run
      # (e.g. two ranks sharing the single GPU of a test box)
      backend = os.environ.get('EPOS_DIST_BACKEND') or (
          'nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
      torch.cuda.set_device(local_rank)
      dist.init_process_group(backend, rank=rank, world_size=world,
                              device_id=torch.device('cuda', local_rank))
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)
  return rank, world, local_rank


def shard_range(n_items, rank, world):
  """Rank r takes items [r*n/G, (r+1)*n/G); remainders go to the
  first ranks. Returns (begin, end)."""
  base, rem = divmod(n_items, world)
  begin = rank * base + min(rank, rem)
  return begin, begin + base + (1 if rank < rem else 0)


def poses_to_records(poses, max_records):
  """List of pose dicts (infer.py:496-503) -> (records f64[max_records,17], n)."""
  rec = np.zeros((max_records, RECORD_COLS), np.float64)
  n = min(len(poses), max_records)
  for i in range(n):
    p = poses[i]
    rec[i, 0] = p['scene_id']
    rec[i, 1] = p['im_id']
    rec[i, 2] = p['obj_id']
    rec[i, 3] = p['score']
    rec[i, 4:13] = np.asarray(p['R']).reshape(9)
    rec[i, 13:16] = np.asarray(p['t']).reshape(3)
    rec[i, 16] = p.get('time', -1.0)
  return rec, n


def records_to_poses(rec, n):
  out = []
  for i in range(n):
    r = rec[i]
    out.append({'scene_id': int(r[0]), 'im_id': int(r[1]), 'obj_id': int(r[2]),
                'score': float(r[3]), 'R': r[4:13].reshape(3, 3).copy(),
                't': r[13:16].reshape(3, 1).copy(), 'time': float(r[16])})
  return out


def gather_poses(poses, max_records=None, device=None):
  """All ranks contribute their pose lists; every rank gets the concatenation in
  rank order (rank 0 writes the CSV). One all_gather of a fixed-size tensor:
  [max_records*17 + 1] float64 per rank -- KB-sized, latency bound. ``max_records``
  must be the same on every rank (poses beyond it are dropped); None = agreed on
  by a MAX all-reduce of the local counts."""
  world = dist.get_world_size() if dist.is_initialized() else 1
  if world == 1:
    return list(poses)
  if device is None:
    device = (torch.device('cuda', torch.cuda.current_device())
              if dist.get_backend() == 'nccl' else torch.device('cpu'))
  if max_records is None:
    # every rank must bring the SAME tensor size to all_gather_into_tensor: agree on
    # the largest local count first (one 8-byte MAX all-reduce)
    max_records = max(1, int(max_over_ranks(float(len(poses)), device)))
  rec, n = poses_to_records(poses, max_records)
  local = torch.empty(max_records * RECORD_COLS + 1, dtype=torch.float64,
                      device=device)
  local[:-1] = torch.from_numpy(rec.reshape(-1)).to(device)
  local[-1] = n
  out = torch.empty(world * local.numel(), dtype=torch.float64, device=device)
  dist.all_gather_into_tensor(out, local)
  out = out.cpu().numpy().reshape(world, -1)
  merged = []
  for r in range(world):
    cnt = int(out[r, -1])
    merged += records_to_poses(out[r, :-1].reshape(max_records, RECORD_COLS), cnt)
  return merged


def max_over_ranks(value, device=None):
  """MAX all-reduce of a python float (bench timing contract)."""
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return value
  if device is None:
    device = (torch.device('cuda', torch.cuda.current_device())
              if dist.get_backend() == 'nccl' else torch.device('cpu'))
  t = torch.tensor([value], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())


def barrier():
  if dist.is_initialized() and dist.get_world_size() > 1:
    dist.barrier()
