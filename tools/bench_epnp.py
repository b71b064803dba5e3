"""Times the OpenCV fitting method (epos_solve_pnp_ransac_device) alone: S slots of N
correspondences, 400 iterations, HIP events around the three launches.
  python tools/bench_epnp.py            ->  gpurun_out/... (stdout)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epos_amd import _lib                              # noqa: E402
from tests.test_oracle_epnp import scene, K            # noqa: E402


def main():
  lib = _lib.load()
  d = 'cuda:0'
  p = _lib.PnpRansacParams()
  lib.epos_pnp_ransac_params_default(ctypes.byref(p))
  vp = lambda t: ctypes.c_void_p(t.data_ptr())
  for S, n, outl in [(1, 500, 0.5), (5, 2000, 0.5), (5, 2000, 0.0), (5, 10000, 0.5), (32, 2000, 0.5)]:
    sc = [scene(10 * i + n, n, sigma=1.0, outliers=outl) for i in range(S)]
    xy = torch.from_numpy(np.concatenate([s[1] for s in sc])).to(d)
    xyz = torch.from_numpy(np.concatenate([s[0] for s in sc])).to(d)
    base = torch.arange(S + 1, dtype=torch.int64, device=d) * n
    Ks = torch.from_numpy(np.tile(K.reshape(1, 9), (S, 1))).to(d)
    N = S * n
    work = torch.empty(lib.epos_pnp_ransac_workspace_bytes(S, N, ctypes.byref(p)),
                       dtype=torch.uint8, device=d)
    poses = torch.zeros(S, 12, dtype=torch.float64, device=d)
    succ = torch.zeros(S, dtype=torch.int32, device=d)
    mask = torch.zeros(N, dtype=torch.uint8, device=d)
    info = torch.zeros(S, 4, dtype=torch.int32, device=d)

    def run():
      _lib.check(lib.epos_solve_pnp_ransac_device(
          vp(xy), vp(xyz), vp(base), S, N, vp(Ks), ctypes.byref(p), vp(work), vp(poses),
          vp(succ), vp(mask), vp(info), None), 'run')
    for _ in range(3):
      run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
      run()
    e1.record()
    torch.cuda.synchronize()
    print('S=%2d N=%5d outliers=%.1f: %.1f us per call, found %d/%d, sets evaluated %s' % (
        S, n, outl, e0.elapsed_time(e1) * 1000 / 20, int(succ.sum()), S,
        info[:, 3].cpu().numpy().tolist()[:5]))


if __name__ == '__main__':
  main()
