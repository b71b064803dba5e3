"""Many-to-many 2D-3D correspondences on MI355X -- the operator that replaces
``epos_lib/corresp.py::establish_many_to_many`` (corresp.py:9-101).

``establish_many_to_many`` keeps the reference's signature, argument meaning,
output dict and dtypes (px_id/frag_id int64, coord_2d/coord_3d float64,
conf* float32); ``CorrExtractor`` is the batched, device-resident form used by
the end-to-end pipeline (its outputs feed the PnP-RANSAC kernels without a host
round trip). Both run the same three HIP launches (csrc/corresp.hip).
"""
import ctypes

import numpy as np
import torch

from epos_amd import _lib


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr())


def _stream(dev):
  return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def pack_model_store(model_store, num_objs, num_frags):
  """frag_centers / frag_sizes dicts (datagen.py:93-124) -> dense f64 arrays
  [O,F,3] / [O,F] indexed by obj_id - 1 (objects without a model stay zero)."""
  centers = np.zeros((num_objs, num_frags, 3), np.float64)
  sizes = np.zeros((num_objs, num_frags), np.float64)
  for obj_id in model_store.dp_model['obj_ids']:
    if 1 <= obj_id <= num_objs:
      centers[obj_id - 1] = np.asarray(model_store.frag_centers[obj_id],
                                       np.float64)
      sizes[obj_id - 1] = np.asarray(model_store.frag_sizes[obj_id], np.float64)
  return centers, sizes


def check_obj_ids(obj_ids, num_objs):
  """Every object the kernels are asked about must have head channels: they index
  obj_confs[..., obj_id], frag_confs[..., obj_id - 1, :] and the fragment tables at
  obj_id - 1 (corresp.py:46,60,76 -- the reference raises IndexError there, e.g. for
  an LM model store used with an LM-O checkpoint)."""
  bad = [int(o) for o in obj_ids if not 1 <= int(o) <= num_objs]
  if bad:
    raise ValueError('object ids %s have no channels in a %d-object network '
                     '(valid: 1..%d)' % (bad, num_objs, num_objs))


class CorrExtractor(object):
  """Batched device-side extractor with preallocated buffers.

  capacity = maximum number of correspondences (rows) per call over all slots.
  """

  def __init__(self, batch, out_h, out_w, num_objs, num_frags, frag_centers,
               frag_sizes, max_slots, capacity, device='cuda:0'):
    if not torch.cuda.is_available():
      raise _lib.EposError('CorrExtractor needs a HIP device (no CPU fallback).')
    self.lib = _lib.load()
    self.dev = torch.device(device)
    self.B, self.h, self.w = batch, out_h, out_w
    self.P = out_h * out_w
    self.O, self.F = num_objs, num_frags
    self.max_slots, self.capacity = max_slots, capacity
    d = self.dev
    self.centers = torch.from_numpy(np.ascontiguousarray(frag_centers)).to(d)
    self.sizes = torch.from_numpy(np.ascontiguousarray(frag_sizes)).to(d)
    S, P = max_slots, self.P
    self.slots = torch.zeros(S, 2, dtype=torch.int32, device=d)
    self.px_off = torch.empty(S, P, dtype=torch.int32, device=d)
    self.corr_off = torch.empty(S, P, dtype=torch.int32, device=d)
    self.frag_mask = torch.empty(S, P, dtype=torch.int64, device=d)
    self.totals = torch.zeros(S, 2, dtype=torch.int32, device=d)
    self.slot_base = torch.zeros(S + 1, dtype=torch.int64, device=d)
    self.overflow = torch.zeros(1, dtype=torch.int32, device=d)
    self._alloc_out(capacity)

  def _alloc_out(self, capacity):
    d = self.dev
    self.capacity = capacity
    n = max(capacity, 1)
    self.px_id = torch.empty(n, dtype=torch.int64, device=d)
    self.frag_id = torch.empty(n, dtype=torch.int64, device=d)
    self.coord_2d = torch.empty(n, 2, dtype=torch.float64, device=d)
    self.coord_3d = torch.empty(n, 3, dtype=torch.float64, device=d)
    self.conf = torch.empty(n, dtype=torch.float32, device=d)
    self.conf_obj = torch.empty(n, dtype=torch.float32, device=d)
    self.conf_frag = torch.empty(n, dtype=torch.float32, device=d)
    self._out = _lib.CorrOut(
        px_id=_ptr(self.px_id), frag_id=_ptr(self.frag_id),
        coord_2d=_ptr(self.coord_2d), coord_3d=_ptr(self.coord_3d),
        conf=_ptr(self.conf), conf_obj=_ptr(self.conf_obj),
        conf_frag=_ptr(self.conf_frag))

  def set_slots(self, slots):
    """slots: list of (image index, obj_id)."""
    if len(slots) > self.max_slots:
      raise ValueError('too many slots (%d > %d)' % (len(slots), self.max_slots))
    check_obj_ids([o for _, o in slots], self.O)
    if any(not 0 <= im < self.B for im, _ in slots):
      raise ValueError('slot image index outside the batch of %d' % self.B)
    self.S = len(slots)
    if self.S:
      arr = torch.tensor(slots, dtype=torch.int32).reshape(-1, 2)
      self.slots[:self.S].copy_(arr, non_blocking=True)

  def count(self, obj_confs, frag_confs, min_obj_conf, min_frag_rel_conf):
    """Launches mask + scan + slot bases; nothing synchronises."""
    if not self.S:
      return
    s = _stream(self.dev)
    _lib.check(self.lib.epos_corr_count(
        _ptr(obj_confs), _ptr(frag_confs), _ptr(self.slots), self.S, self.B,
        self.P, self.O, self.F, float(np.float32(min_obj_conf)),
        float(np.float32(min_frag_rel_conf)), _ptr(self.px_off),
        _ptr(self.corr_off), _ptr(self.frag_mask), _ptr(self.totals), s),
               'epos_corr_count')
    _lib.check(self.lib.epos_corr_slot_bases(
        _ptr(self.totals), self.S, _ptr(self.slot_base), s),
               'epos_corr_slot_bases')

  def fill(self, obj_confs, frag_confs, frag_coords, output_scale):
    if not self.S:
      return
    self.overflow.zero_()
    _lib.check(self.lib.epos_corr_fill(
        _ptr(obj_confs), _ptr(frag_confs), _ptr(frag_coords),
        _ptr(self.centers), _ptr(self.sizes), _ptr(self.slots), self.S, self.B,
        self.P, self.w, self.O, self.F, 1.0 / output_scale, _ptr(self.px_off),
        _ptr(self.corr_off), _ptr(self.frag_mask), _ptr(self.slot_base),
        self.capacity, ctypes.byref(self._out), _ptr(self.overflow),
        _stream(self.dev)), 'epos_corr_fill')


def establish_many_to_many(
      obj_confs, frag_confs, frag_coords, gt_obj_ids, model_store, output_scale,
      min_obj_conf, min_frag_rel_conf, project_to_surface, only_annotated_objs,
      device='cuda:0'):
  """Drop-in for corresp.py:9-33 (same arguments, same result dict).

  obj_confs [h,w,O+1], frag_confs [h,w,O,F], frag_coords [h,w,O,F,3]: numpy
  arrays (copied to HBM) or torch tensors already on the device.
  """
  if project_to_surface and not getattr(model_store, 'models', None):
    raise ValueError(
        'project_to_surface needs model_store.models = {obj_id: {"pts", "faces"}} '
        '(datagen.py:68-84; epos_amd.ply.load_models)')
  dev = torch.device(device)

  def to_dev(a):
    t = torch.as_tensor(a)
    return t.to(device=dev, dtype=torch.float32).contiguous()
  obj_confs, frag_confs, frag_coords = map(to_dev, (obj_confs, frag_confs,
                                                    frag_coords))
  h, w, o1 = obj_confs.shape
  num_objs, num_frags = frag_confs.shape[2], frag_confs.shape[3]
  assert o1 == num_objs + 1
  obj_ids = [o for o in model_store.dp_model['obj_ids']
             if not (only_annotated_objs and o not in gt_obj_ids)]  # :39-43
  if not obj_ids:
    return {}
  check_obj_ids(obj_ids, num_objs)
  centers, sizes = pack_model_store(model_store, num_objs, num_frags)
  ex = CorrExtractor(1, h, w, num_objs, num_frags, centers, sizes,
                     max_slots=len(obj_ids), capacity=0, device=device)
  ex.set_slots([(0, o) for o in obj_ids])
  ex.count(obj_confs, frag_confs, min_obj_conf, min_frag_rel_conf)
  totals = ex.totals[:ex.S].cpu().numpy()          # sync: sizes are data dependent
  base = np.concatenate([[0], np.cumsum(totals[:, 1], dtype=np.int64)])
  ex._alloc_out(int(base[-1]))
  ex.fill(obj_confs, frag_confs, frag_coords, output_scale)
  if int(ex.overflow.item()):
    raise _lib.EposError('correspondence buffer overflow')
  host = {k: getattr(ex, k).cpu().numpy() for k in
          ['px_id', 'frag_id', 'coord_2d', 'coord_3d', 'conf', 'conf_obj',
           'conf_frag']}
  corresp = {}
  for s, obj_id in enumerate(obj_ids):
    if totals[s, 0] == 0:
      continue                                     # no masked pixel: key absent (:49)
    lo, hi = int(base[s]), int(base[s + 1])
    corresp[obj_id] = {k: v[lo:hi].copy() for k, v in host.items()}
    if project_to_surface:                         # corresp.py:87-88
      m = model_store.models[obj_id]
      corresp[obj_id]['coord_3d'] = project_pts_to_model(
          ex.coord_3d[lo:hi], m['pts'], m['faces'], device=device)
  return corresp


def project_pts_to_model(pts, verts, faces, device='cuda:0', return_faces=False):
  """ObjectModelStore.project_pts_to_model (datagen.py:128-154): closest point of the
  triangle mesh (verts [V,3], faces [F,3]) for every row of pts [N,3]. pts may be a
  device tensor (f64) or a host array; returns a host float64 array."""
  lib = _lib.load()
  dev = torch.device(device)
  P = torch.as_tensor(pts).to(device=dev, dtype=torch.float64).contiguous()
  V = torch.as_tensor(np.ascontiguousarray(verts, np.float64)).to(dev)
  Fc = torch.as_tensor(np.ascontiguousarray(faces, np.int32)).to(dev)
  out = torch.empty_like(P)
  fidx = torch.empty(P.shape[0], dtype=torch.int32, device=dev)
  _lib.check(lib.epos_project_to_mesh_f64(
      _ptr(P), P.shape[0], _ptr(V), V.shape[0], _ptr(Fc), Fc.shape[0], _ptr(out),
      _ptr(fidx), _stream(dev)), 'epos_project_to_mesh_f64')
  if return_faces:
    return out.cpu().numpy(), fidx.cpu().numpy()
  return out.cpu().numpy()
