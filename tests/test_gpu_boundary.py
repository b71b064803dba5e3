"""GPU tests of the boundary's contract as the HEADER states it (include/epos_hip.h): a
maintainer who binds from that text alone must get right poses -- for as many instances as
the frame asks for (no clamp to 4)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import fit_scenes as fs   # noqa: E402


def _six_instances(rng):
  """Six instances of one object on a 3 x 2 grid in front of the camera (well separated
  in the image, so each explains its own pixels)."""
  inst = []
  for i in range(6):
    gx, gy = i % 3, i // 3
    t = np.array([(gx - 1) * 210.0, (gy - 0.5) * 230.0, 1150.0 + 40.0 * i])
    inst.append((fs.rand_rot(rng), t))
  return inst


def test_pose_layout_as_the_header_documents_it_with_six_instances():
  """Bind epos_find6d_poses from the header's words only: instance j = poses[12 j ..
  12 j + 11], R row-major in [0..8] (R[r][c] = poses[12 j + 3 r + c]), t in [9..11]; six
  instances of one object in one call (max_k = 6). Every returned pose must reproject its
  own labelled correspondences within the inlier threshold and match one ground-truth
  instance; a 3x4 reading of the same 12 doubles must NOT (that was the header's old
  text)."""
  from epos_amd import _lib
  lib = _lib.load()
  header = open(os.path.join(ROOT, 'include', 'epos_hip.h')).read()
  doc = header[header.index('Host-pointer drop-in for pyprogressivex.find6DPoses'):]
  doc = doc[:doc.index('int epos_find6d_poses(')]
  assert re.search(r'R row-major in \[0\.\.8\]', doc) and re.search(r't in \[9\.\.11\]', doc)
  assert 'NOT a row-major 3x4' in doc
  rng = np.random.RandomState(11)
  gt = _six_instances(rng)
  xy, xyz, src, kind = fs.dense_scene(rng, gt, sigma3d=0.5, sym=0.0, outlier=0.2)
  n = len(xy)
  prm = _lib.FitParams()
  lib.epos_fit_params_default(ctypes.byref(prm))
  prm.max_model_number = 6
  max_k = 6
  poses = np.zeros(max_k * 12)
  scores = np.zeros(max_k)
  labels = np.zeros(n, np.int32)
  K = np.ascontiguousarray(fs.K_YCBV)
  k = lib.epos_find6d_poses(xy.ctypes.data, xyz.ctypes.data, n, K.ctypes.data,
                            ctypes.byref(prm), 5, poses.ctypes.data, labels.ctypes.data,
                            scores.ctypes.data, max_k)
  assert k == 6, k
  matched = set()
  for j in range(k):
    blk = poses[12 * j:12 * j + 12]
    R = np.array([[blk[3 * r + c] for c in range(3)] for r in range(3)])   # header text
    t = blk[9:12]
    assert abs(np.linalg.det(R) - 1) < 1e-9 and np.allclose(R @ R.T, np.eye(3), atol=1e-9)
    mine = labels == j
    assert mine.sum() >= 50
    res, z = fs.reproj_residuals(R, t, K, xy[mine], xyz[mine])
    assert (z > 0).all() and np.sqrt((res * res).sum(1)).max() < prm.threshold
    errs = [fs.pose_err_sym(R, t, Rg, tg) for Rg, tg in gt]
    best = int(np.argmin([e[0] + e[1] / 10 for e in errs]))
    assert errs[best][0] < 1.0 and errs[best][1] < 0.005 * gt[best][1][2], errs[best]
    matched.add(best)
    # the 3x4 misreading is visibly wrong
    M = blk.reshape(3, 4)
    res34, _ = fs.reproj_residuals(M[:, :3], M[:, 3], K, xy[mine], xyz[mine])
    assert not np.sqrt((res34 * res34).sum(1)).max() < prm.threshold
  assert matched == set(range(6))


def test_pipeline_never_clamps_instance_counts_silently():
  """EposPipeline: a frame asking for more instances of an object than the plan was built
  for raises (infer.py sizes the plan from the frames it reads); 'all found' (detection)
  runs up to the cap and says so when it is reached."""
  import torch
  import warnings
  from epos_amd import _lib, pipeline, synthetic, weights
  O, F, H, W = 2, 64, 96, 128
  ckpt = weights.random_init(num_objs=O, num_frags=F, seed=0, randomize_bn=True,
                             logits_std=1.0)
  store = synthetic.ModelStore(O, F, seed=1)
  pipe = pipeline.EposPipeline(ckpt, 1, H, W, O, F, store, max_instances=2, capacity=1 << 16)
  img = torch.from_numpy(synthetic.image(0, H, W)[None]).cuda()
  Ks = synthetic.YCBV_K[None]
  with pytest.raises(_lib.EposError, match='max_instances=2'):
    pipe.make_slots([{1: 3}])
  slots, wants = pipe.make_slots([{1: 2, 2: 1}])
  assert wants == [2, 1]
  # on_excess='clamp': fit max_instances of them, warn once, cap hits of 'all found' recorded
  pipe.on_excess = 'clamp'
  with warnings.catch_warnings(record=True) as rec:
    warnings.simplefilter('always')
    slots, wants = pipe.make_slots([{1: 3, 2: 1}])
    pipe.make_slots([{1: 5}])
  assert wants == [2, 1] and len([w for w in rec if 'clamped' in str(w.message)]) == 1
  # one warning for the whole run, but every clamped request is on record
  assert pipe.last_clamped == [(0, 1, 5, 2)] and pipe.clamped_count == 2
  assert list(pipe.clamped) == [(0, 1, 3, 2), (0, 1, 5, 2)]
  pipe.on_excess = 'raise'
  slots, wants = pipe.make_slots([{}], task_type='detection')
  assert wants == [-1, -1] and len(slots) == O
  with warnings.catch_warnings(record=True):
    warnings.simplefilter('always')
    poses, _ = pipe.process_batch(img, Ks, [{1: 2, 2: 1}])
  assert all(p['R'].shape == (3, 3) and p['t'].shape == (3, 1) for p in poses)


def test_a_timed_out_hand_off_is_reported_not_hidden(tmp_path):
  """The refits of a slot are summed by four workgroups that meet in global memory; a
  workgroup that waits too long for its siblings gives up (bounded spin). With the bound
  set to ZERO polls (EPOS_FIT_SPIN_MAX=0, read once per process: a subprocess) every
  hand-off that has to wait at all fails -- the call must then raise (num_models == -1 /
  EPOS_E_INTERNAL), never return poses computed from incomplete sums."""
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  script = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import fit_scenes as fs
from epos_amd import fitting, _lib
from oracle import pnp_ref
rng = np.random.RandomState(7)
R = fs.rand_rot(rng); t = np.array([10.0, -20.0, 750.0])
xy, xyz, _, _ = fs.dense_scene(rng, [(R, t)], sigma3d=1.0, sym=0.3, outlier=0.3)
raised = 0
for seed in range(6):
  try:
    got = fitting.find6DPoses(xy, xyz, fs.K_YCBV, seed=seed)
  except _lib.EposError as e:
    assert 'timed out' in str(e), str(e)
    raised += 1
    continue
  ref = pnp_ref.find6DPoses(xy, xyz, fs.K_YCBV, seed=seed)     # no time-out happened:
  assert np.array_equal(got[1], ref[1])                          # then the result is right
print('RAISED', raised)
''' % (root, os.path.join(root, 'tests'))
  r = subprocess.run([sys.executable, '-c', script], env=dict(os.environ, EPOS_FIT_SPIN_MAX='0'),
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stdout + r.stderr
  assert int(r.stdout.strip().split()[-1]) >= 1, r.stdout
