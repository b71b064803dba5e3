"""The neighbour lists of the fitting stage (round 3: the 5-D neighbourhood graph is built
once per call and walked by the spatial-coherence sweeps of every round and by the joint
refinement) against the window scans they replace (EPOS_FIT_NB_LISTS=0, read once per
process: a subprocess per mode). Labels, poses and scores must agree BIT FOR BIT -- the sums
are integers -- on a sparse scene (lists fit), on a dense many-to-many scene (the lists of at
least some tiles overflow their caps: that slot falls back to scanning) and on a two-instance
scene (multi-round search + joint refinement)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import fit_scenes as fs
from epos_amd import fitting
out = {}
rng = np.random.RandomState(5)
scenes = {
  'sparse': ([(fs.rand_rot(rng), np.array([20.0, -10.0, 800.0]))], dict(sigma3d=0.5, sym=0.0, outlier=0.1), 1),
  'dense': ([(fs.rand_rot(rng), np.array([-30.0, 15.0, 520.0]))], dict(sigma3d=0.5, sym=1.0, outlier=1.0), 1),
  'two': ([(fs.rand_rot(rng), np.array([-38.0, 10.0, 760.0])), (fs.rand_rot(rng), np.array([42.0, -5.0, 790.0]))],
          dict(sigma3d=0.7, sym=1.0, outlier=0.2), 3),
}
for name, (inst, kw, k) in scenes.items():
  xy, xyz, src, kind = fs.dense_scene(np.random.RandomState(9), inst, **kw)
  P, lab, sc = fitting.find6DPoses(xy, xyz, fs.K_YCBV, max_model_number=k, seed=3)
  out[name + '_P'] = np.zeros(0) if P is None else P
  out[name + '_lab'] = lab
  out[name + '_sc'] = np.zeros(0) if sc is None else np.asarray(sc)
  out[name + '_n'] = np.array([len(xy)])
np.savez(sys.argv[1], **out)
'''


def test_neighbour_lists_equal_window_scans(tmp_path):
  """Five builds of the same labelling: window scans with the candidates in scalar registers
  followed by delta sweeps over the flipped candidates (the default), full scans every sweep
  (EPOS_FIT_DELTA=0), the LDS-staged scans they replaced (EPOS_FIT_SCAN=0), with and without
  the neighbour lists."""
  res = {}
  # (scan, lists, delta): delta = the sweeps after the first only revisit flipped candidates
  modes = [('1', '1', '1'), ('1', '1', '0'), ('1', '0', '1'), ('0', '1', '1'), ('0', '0', '1')]
  for mode in modes:
    path = str(tmp_path / ('fit_%s%s%s.npz' % mode))
    r = subprocess.run([sys.executable, '-c', SCRIPT % (ROOT, os.path.join(ROOT, 'tests')), path],
                       env=dict(os.environ, EPOS_FIT_SCAN=mode[0], EPOS_FIT_NB_LISTS=mode[1],
                                EPOS_FIT_DELTA=mode[2]),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    res[mode] = dict(np.load(path))
  ref = res[modes[0]]
  for mode in modes[1:]:
    for k in ref:
      assert np.array_equal(ref[k], res[mode][k]), (mode, k)
  assert ref['sparse_P'].size and ref['dense_P'].size and ref['two_P'].shape[0] >= 6
  assert ref['dense_n'][0] > 2 * ref['sparse_n'][0]
