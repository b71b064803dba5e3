"""End-to-end EPOS inference on one MI355X: network -> correspondences ->
PnP-RANSAC, everything resident in HBM, one host synchronisation per batch.

Mirrors ``process_image`` of scripts/infer.py:348-554 for a batch of images
(the reference is hard-wired to batch 1, infer.py:610): same stage split
(prediction / establish_corr / fitting, infer.py:372-374,395,407,410,534-535),
same pose records (infer.py:496-503).

What changed relative to the reference's data flow: the three
dense head tensors (~415 MB at YCB-V) are never copied to the host; the
correspondence kernels read them in place and write pooled f64 arrays that the
RANSAC kernels consume directly; all objects of all images are fitted by one
launch sequence instead of a serial per-object loop (infer.py:412).
"""
import ctypes

import numpy as np
import torch

from epos_amd import _lib
from epos_amd import corresp as _corresp
from epos_amd import fitting as _fitting
from epos_amd import model as _model
from epos_amd import weights as W

LOCALIZATION = 'localization'   # common.py
DETECTION = 'detection'


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr())


class EposPipeline(object):

  def __init__(self, checkpoint, batch, height, width, num_objs, num_frags,
               model_store, fit_params=None, corr_min_obj_conf=0.1,
               corr_min_frag_rel_conf=0.5, max_slots=None, capacity=1 << 20,
               max_instances=4, model_options=None, device='cuda:0',
               use_graph=True):
    self.lib = _lib.load()
    self.dev = torch.device(device)
    self.B, self.H, self.W = batch, height, width
    self.O, self.F = num_objs, num_frags
    self.net = _model.get_net(checkpoint, batch, height, width, num_objs,
                              num_frags, model_options, device)
    self.use_graph = use_graph
    self.output_scale = 1.0 / 4            # decoder output stride 4 (infer.py:586-591)
    self.tau_a, self.tau_b = corr_min_obj_conf, corr_min_frag_rel_conf
    self.max_slots = max_slots or batch * num_objs
    self.max_k = max_instances
    centers, sizes = _corresp.pack_model_store(model_store, num_objs, num_frags)
    self.obj_ids = list(model_store.dp_model['obj_ids'])
    self.corr = _corresp.CorrExtractor(
        batch, self.net.out_h, self.net.out_w, num_objs, num_frags, centers,
        sizes, self.max_slots, capacity, device)
    self.fit = fit_params or _fitting.fit_params()
    S = self.max_slots
    d = self.dev
    wbytes = self.lib.epos_fit_workspace_bytes(S, capacity, ctypes.byref(self.fit),
                                               self.max_k)
    if wbytes < 0:
      raise _lib.EposError('epos_fit_workspace_bytes failed')
    self.work = torch.empty(wbytes, dtype=torch.uint8, device=d)
    self.Ks = torch.zeros(S, 9, dtype=torch.float64, device=d)
    self.max_models = torch.zeros(S, dtype=torch.int32, device=d)
    self.seeds = torch.zeros(S, dtype=torch.int64, device=d)
    self.poses = torch.zeros(S, self.max_k, 12, dtype=torch.float64, device=d)
    self.scores = torch.zeros(S, self.max_k, dtype=torch.float64, device=d)
    self.num_models = torch.zeros(S, dtype=torch.int32, device=d)
    self.labels = torch.empty(max(capacity, 1), dtype=torch.int32, device=d)
    # pinned host mirror for the single D2H copy of the results
    self._ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

  # --------------------------------------------------------------------
  def make_slots(self, targets, task_type=LOCALIZATION):
    """targets: per image, dict obj_id -> number of instances (localization:
    the GT instance counts, infer.py:383-392,462-463). Detection: every object of
    the model store, unlimited instances (infer.py:464-465)."""
    slots, wants = [], []
    for im, t in enumerate(targets):
      for obj_id in self.obj_ids:                      # corresp.py:39 order
        if task_type == LOCALIZATION:
          if obj_id not in t:                          # corresp.py:42-43
            continue
          wants.append(min(int(t[obj_id]), self.max_k))
        else:
          wants.append(-1)
        slots.append((im, obj_id))
    return slots, wants

  def process_batch(self, images, Ks, targets, task_type=LOCALIZATION,
                    image_ids=None, scene_ids=None, seed=0, timing=False):
    """images f32 [B,H,W,3]; Ks [B,3,3]; targets per image {obj_id: count}.
    Returns (poses, run_times) like process_image (infer.py:348-554)."""
    B = self.B
    slots, wants = self.make_slots(targets, task_type)
    S = len(slots)
    if timing:
      self._ev[0].record()
    pred = self.net.forward(images, use_graph=self.use_graph)
    if timing:
      self._ev[1].record()
    poses_out = []
    if S:
      Ks = np.asarray(Ks, np.float64).reshape(B, 9)
      self.corr.set_slots(slots)
      self.Ks[:S].copy_(torch.from_numpy(Ks[[s[0] for s in slots]]),
                        non_blocking=True)
      self.max_models[:S].copy_(torch.tensor(wants, dtype=torch.int32),
                                non_blocking=True)
      sd = [(seed * 1000003 + (image_ids[im] if image_ids is not None else im)
             * 1009 + obj) & 0x7fffffffffffffff for im, obj in slots]
      self.seeds[:S].copy_(torch.tensor(sd, dtype=torch.int64),
                           non_blocking=True)
      self.corr.count(pred[W.PRED_OBJ_CONF], pred[W.PRED_FRAG_CONF],
                      self.tau_a, self.tau_b)
      self.corr.fill(pred[W.PRED_OBJ_CONF], pred[W.PRED_FRAG_CONF],
                     pred[W.PRED_FRAG_LOC], self.output_scale)
      if timing:
        self._ev[2].record()
      max_k = self.max_k if task_type != LOCALIZATION else max(1, max(wants))
      _lib.check(self.lib.epos_find6d_poses_device(
          _ptr(self.corr.coord_2d), _ptr(self.corr.coord_3d),
          _ptr(self.corr.slot_base), S, self.corr.capacity, _ptr(self.Ks),
          _ptr(self.max_models), _ptr(self.seeds), ctypes.byref(self.fit),
          max_k, _ptr(self.work), _ptr(self.poses), _ptr(self.scores),
          _ptr(self.num_models), _ptr(self.labels),
          ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)),
                 'epos_find6d_poses_device')
      # NB: poses / scores are laid out [S, max_k(call), ...] for this call.
      if timing:
        self._ev[3].record()
      nm = self.num_models[:S].cpu().numpy()          # the one synchronisation
      if int(self.corr.overflow.item()):
        raise _lib.EposError(
            'correspondence capacity (%d rows) exceeded' % self.corr.capacity)
      ph = self.poses.view(-1)[:S * max_k * 12].cpu().numpy().reshape(S, max_k, 12)
      sh = self.scores.view(-1)[:S * max_k].cpu().numpy().reshape(S, max_k)
      for s, (im, obj_id) in enumerate(slots):
        for i in range(int(nm[s])):
          poses_out.append({
              'scene_id': scene_ids[im] if scene_ids is not None else 0,
              'im_id': image_ids[im] if image_ids is not None else im,
              'obj_id': obj_id,
              'R': ph[s, i, :9].reshape(3, 3).copy(),
              't': ph[s, i, 9:].reshape(3, 1).copy(),
              'score': float(sh[s, i]),
          })
    else:
      torch.cuda.synchronize(self.dev)
    run_times = {}
    if timing and S:
      torch.cuda.synchronize(self.dev)
      e = self._ev
      run_times = {'prediction': e[0].elapsed_time(e[1]) * 1e-3,
                   'establish_corr': e[1].elapsed_time(e[2]) * 1e-3,
                   'fitting': e[2].elapsed_time(e[3]) * 1e-3}
      run_times['total'] = sum(run_times.values())
      for p in poses_out:
        p['time'] = run_times['total']
    return poses_out, run_times
