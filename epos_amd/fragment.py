"""Surface fragmentation of the object models on MI355X -- mirrors
``epos_lib/fragment.py::fragmentation_fps`` (fragment.py:8-54) and the size rule
of ``ObjectModelStore.fragment_models`` (datagen.py:86-126); writes / reads the
reference's ``fragments.pkl`` (datagen.py:254-296)."""
import ctypes
import pickle

import numpy as np
import torch

from epos_amd import _lib


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr())


def fragmentation_fps(vertices, num_frags, device='cuda:0'):
  """Same contract as fragment.py:8-54: returns ([num_frags,3] f64 fragment
  centres, [num_vertices] int64 fragment id per vertex). Runs the HIP kernels of
  csrc/fragment.hip."""
  if not torch.cuda.is_available():
    raise _lib.EposError('fragmentation_fps needs a HIP device (no CPU fallback).')
  lib = _lib.load()
  v = torch.as_tensor(np.ascontiguousarray(vertices, np.float64)).to(device)
  n = v.shape[0]
  assert n >= num_frags                                    # datagen.py:106
  nn = torch.empty(n, dtype=torch.float64, device=device)
  centers = torch.empty(num_frags, 3, dtype=torch.float64, device=device)
  cidx = torch.empty(num_frags, dtype=torch.int32, device=device)
  ids = torch.empty(n, dtype=torch.int32, device=device)
  stream = ctypes.c_void_p(torch.cuda.current_stream(v.device).cuda_stream)
  _lib.check(lib.epos_fragmentation_fps(_ptr(v), n, num_frags, _ptr(nn),
                                        _ptr(centers), _ptr(cidx), _ptr(ids),
                                        stream), 'epos_fragmentation_fps')
  return centers.cpu().numpy(), ids.cpu().numpy().astype(np.int64)


def fragment_models(models_pts, num_frags, device='cuda:0'):
  """datagen.py:86-126: models_pts {obj_id: [V,3] vertices (mm)} ->
  (frag_centers {obj_id: f64[F,3]}, frag_sizes {obj_id: f64[F]}); the size of a
  fragment is the longest side of its bounding box, at least 5 mm."""
  frag_centers, frag_sizes = {}, {}
  for obj_id, pts in models_pts.items():
    pts = np.asarray(pts, np.float64)
    if num_frags == 1:                                     # datagen.py:97-102
      frag_centers[obj_id] = np.array([[0., 0., 0.]])
      ids = np.zeros(pts.shape[0], np.int64)
    else:
      frag_centers[obj_id], ids = fragmentation_fps(pts, num_frags, device)
    sizes = []
    for f in range(num_frags):
      fp = pts[ids == f]
      bb = np.max(fp, axis=0) - np.min(fp, axis=0)
      sizes.append(max(np.max(bb), 5.0))                   # min_frag_size = 5 mm
    frag_sizes[obj_id] = np.array(sizes)
  return frag_centers, frag_sizes


def save_fragments(path, frag_centers, frag_sizes):
  """fragments.pkl as the reference writes it (datagen.py:291-296)."""
  with open(path, 'wb') as f:
    pickle.dump({'frag_centers': frag_centers, 'frag_sizes': frag_sizes}, f,
                protocol=pickle.HIGHEST_PROTOCOL)
