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
import os

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


_STREAMS = {}


def _pipeline_stream(dev, instance):
  """The HIP stream of pipeline number `instance` on a device: created once per process and
  handed out again when a pipeline with that number is built anew. Streams land on the
  runtime's hardware queues in creation order: the first four a process creates get queues of
  their own (GPU_MAX_HW_QUEUES=8), a second set of four ran 0.29 ms per step slower (2.06 ->
  2.36 ms; some of them share a queue), a third set as fast as the first
  (profiles/r06/stream_sets_and_hw_queues.txt). A process that rebuilds its pipelines therefore
  keeps the streams it had."""
  key = (str(dev), int(instance))
  if key not in _STREAMS:
    _STREAMS[key] = torch.cuda.Stream(dev)
  return _STREAMS[key]


class EposPipeline(object):

  def __init__(self, checkpoint, batch, height, width, num_objs, num_frags,
               model_store, fit_params=None, corr_min_obj_conf=0.1,
               corr_min_frag_rel_conf=0.5, max_slots=None, capacity=1 << 20,
               max_instances=4, model_options=None, device='cuda:0',
               use_graph=True, instance=0, sparse_heads=False,
               fitting_method='progressive_x', on_excess='raise', queue=1):
    """on_excess: what launch() does with a frame that asks for more instances of an object
    than `max_instances` (localization): 'raise' (default: EposError BEFORE anything of that
    batch is enqueued -- batches already in flight on other pipelines are unaffected and can
    still be collected) or 'clamp' (fit `max_instances` of them and warn once)."""
    if on_excess not in ('raise', 'clamp'):
      raise ValueError("on_excess must be 'raise' or 'clamp'")
    # queue: how many batches launch() accepts before a collect() (round 6). With 2 the
    # caller enqueues a pipeline's NEXT batch while the current one still runs, so the
    # pipeline's stream never waits for the host between two batches (collect -> bookkeeping
    # -> next frames -> launch: 0.3-0.5 ms per step otherwise). Batches of one pipeline run in
    # stream order on the same device buffers; only the host-side staging (metadata, results,
    # events) exists `queue` times. Readers of the plan's device buffers after collect()
    # (net.outputs(), --vis, --save_corresp) need queue=1.
    if queue < 1:
      raise ValueError('queue must be >= 1')
    self.queue = int(queue)
    self.on_excess = on_excess
    self.lib = _lib.load()
    # 'progressive_x' (infer.py:446-503) or 'opencv_ransac' (infer.py:505-528: one
    # cv2.solvePnPRansac(EPNP) per object, score 0.0) -- common.py:30-31
    if fitting_method not in ('progressive_x', 'opencv_ransac'):
      raise ValueError('Unknown pose fitting method ({}).'.format(fitting_method))
    self.fitting_method = fitting_method
    self.dev = torch.device(device)
    self.B, self.H, self.W = batch, height, width
    self.O, self.F = num_objs, num_frags
    self.net = _model.get_net(checkpoint, batch, height, width, num_objs,
                              num_frags, model_options, device, instance)
    self.use_graph = use_graph
    # sparse_heads: evaluate the fragment heads only for the (image, target
    # object) slots of the batch instead of all O objects. Identical poses (the
    # correspondence stage never reads another object's channels,
    # corresp.py:42-43); the dense prediction dict is then NOT available.
    self.sparse_heads = sparse_heads
    self.output_scale = 1.0 / 4            # decoder output stride 4 (infer.py:586-591)
    self.tau_a, self.tau_b = corr_min_obj_conf, corr_min_frag_rel_conf
    self.max_slots = max_slots or batch * num_objs
    self.max_k = max_instances
    self._warned_cap = False
    self._warned_clamp = False
    # (scene, image, obj, instances) at the cap: last_cap_hits = the batch just collected;
    # cap_hits = the most recent CAP_HITS_KEPT of the run (bounded: a long-running service
    # must not grow), cap_hit_count = how many there were. Requests that on_excess='clamp'
    # cut down are recorded the same way: last_clamped / clamped / clamped_count hold
    # (image index in the batch, obj, requested, fitted).
    import collections
    self.cap_hits = collections.deque(maxlen=self.CAP_HITS_KEPT)
    self.last_cap_hits, self.cap_hit_count = [], 0
    self.clamped = collections.deque(maxlen=self.CAP_HITS_KEPT)
    self.last_clamped, self.clamped_count = [], 0
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
    if fitting_method == 'opencv_ransac':
      self.cv_params = _lib.PnpRansacParams()
      self.lib.epos_pnp_ransac_params_default(ctypes.byref(self.cv_params))
      self.cv_params.iterations_count = self.fit.max_iters      # infer.py:515
      self.cv_params.reprojection_error = self.fit.threshold    # infer.py:516
      self.cv_params.confidence = 0.99                          # infer.py:517
      self.cv_params.min_point_number = self.fit.min_point_number   # infer.py:420-422
      wbytes = max(wbytes, self.lib.epos_pnp_ransac_workspace_bytes(
          S, capacity, ctypes.byref(self.cv_params)))
    self.work = torch.empty(wbytes, dtype=torch.uint8, device=d)
    self.labels = torch.empty(max(capacity, 1), dtype=torch.int32, device=d)
    # Per-step metadata goes up in ONE host->device copy and the results come
    # back in ONE device->host copy (pinned memory on the host side): typed views
    # of two byte buffers.
    K = self.max_k
    self._meta_layout = self._layout([('Ks', 'f8', S * 9), ('seeds', 'i8', S),
                                      ('slots', 'i4', S * 2),
                                      ('max_models', 'i4', S)])
    self._res_layout = self._layout([('poses', 'f8', S * K * 12),
                                     ('scores', 'f8', S * K),
                                     ('num_models', 'i4', S),
                                     ('totals', 'i4', S * 2),
                                     ('overflow', 'i4', 1)])
    self.meta_dev = torch.zeros(self._meta_layout['_size'], dtype=torch.uint8,
                                device=d)
    self._meta_hosts = [torch.zeros(self._meta_layout['_size'], dtype=torch.uint8).pin_memory()
                        for _ in range(self.queue)]
    self.res_dev = torch.zeros(self._res_layout['_size'], dtype=torch.uint8,
                               device=d)
    self._res_hosts = [torch.zeros(self._res_layout['_size'], dtype=torch.uint8).pin_memory()
                       for _ in range(self.queue)]
    mv = lambda buf, lay, k: self._view(buf, lay, k)     # noqa: E731
    self.Ks = mv(self.meta_dev, self._meta_layout, 'Ks')
    self.seeds = mv(self.meta_dev, self._meta_layout, 'seeds')
    self.max_models = mv(self.meta_dev, self._meta_layout, 'max_models')
    self.corr.slots = mv(self.meta_dev, self._meta_layout, 'slots').view(S, 2)
    self.poses = mv(self.res_dev, self._res_layout, 'poses')
    self.scores = mv(self.res_dev, self._res_layout, 'scores')
    self.num_models = mv(self.res_dev, self._res_layout, 'num_models')
    self.corr.totals = mv(self.res_dev, self._res_layout, 'totals').view(S, 2)
    self.corr.overflow = mv(self.res_dev, self._res_layout, 'overflow')
    self.stream = _pipeline_stream(self.dev, instance)
    self._dones = [torch.cuda.Event() for _ in range(self.queue)]
    self._pending = collections.deque()        # launched, not yet collected (oldest first)
    self._launches = 0
    self._evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)]
                 for _ in range(self.queue)]

  _DT = {'f8': (torch.float64, 8), 'i8': (torch.int64, 8), 'i4': (torch.int32, 4)}

  @staticmethod
  def _layout(fields):
    lay, off = {}, 0
    for name, dt, n in fields:
      size = EposPipeline._DT[dt][1] * max(n, 1)
      lay[name] = (off, dt, max(n, 1))
      off = (off + size + 15) // 16 * 16
    lay['_size'] = off
    return lay

  @staticmethod
  def _view(buf, lay, name):
    off, dt, n = lay[name]
    tdt, sz = EposPipeline._DT[dt]
    return buf[off:off + n * sz].view(tdt)

  # --------------------------------------------------------------------
  CAP_HITS_KEPT = 4096

  def make_slots(self, targets, task_type=LOCALIZATION):
    """targets: per image, dict obj_id -> number of instances (localization:
    the GT instance counts, infer.py:383-392,462-463). Detection: every object of
    the model store, unlimited instances (infer.py:464-465)."""
    slots, wants = [], []
    self.last_clamped = []
    for im, t in enumerate(targets):
      for obj_id in self.obj_ids:                      # corresp.py:39 order
        if task_type == LOCALIZATION:
          if obj_id not in t:                          # corresp.py:42-43
            continue
          want = int(t[obj_id])
          if want > self.max_k:
            # never clamp silently (a frame with more instances of one object than the
            # plan was sized for would lose poses): the caller sizes max_instances from
            # the frames it is going to feed (infer.py does), caps the counts itself, or
            # asks for on_excess='clamp'. Raised here, i.e. before anything of this batch
            # is enqueued.
            if self.on_excess == 'raise':
              raise _lib.EposError(
                  'image %d: %d instances of object %d requested, the pipeline was built '
                  'with max_instances=%d' % (im, want, obj_id, self.max_k))
            if not self._warned_clamp:
              import warnings
              warnings.warn('object %d: %d instances requested, clamped to max_instances=%d'
                            % (obj_id, want, self.max_k))
              self._warned_clamp = True
            rec = (im, obj_id, want, self.max_k)
            self.last_clamped.append(rec)
            self.clamped.append(rec)
            self.clamped_count += 1
            want = self.max_k
          wants.append(want)
        else:
          wants.append(-1)                             # all found, up to max_instances
        slots.append((im, obj_id))
    _corresp.check_obj_ids([o for _, o in slots], self.O)
    return slots, wants

  def launch(self, images, Ks, targets, task_type=LOCALIZATION, image_ids=None,
             scene_ids=None, seed=0, timing=False, after_net=None):
    """Enqueues one batch on this pipeline's stream -- network, correspondences,
    PnP-RANSAC and the single device->host copy of the results -- and returns
    without synchronising. ``collect()`` waits for it and builds the pose list.
    Two pipelines used alternately overlap one batch's (latency-bound) fitting
    tail with the next batch's network. after_net: optional callable(pipeline), invoked on
    the pipeline's stream between the network and the correspondence stage (bench.py
    --planted-poses overwrites head values there)."""
    if len(self._pending) >= self.queue:
      raise _lib.EposError('launch() called %s without collect()' % (
          'twice' if self.queue == 1 else '%d times' % (self.queue + 1)))
    q = self._launches % self.queue            # host-side staging of this batch
    ev, mh, res_host, done = self._evs[q], self._meta_hosts[q], self._res_hosts[q], self._dones[q]
    B = self.B
    slots, wants = self.make_slots(targets, task_type)
    S = len(slots)
    max_k = self.max_k if task_type != LOCALIZATION else max([1] + wants)
    if self.fitting_method == 'opencv_ransac':
      max_k = 1                       # "can estimate pose of only one object instance"
    cur = torch.cuda.current_stream(self.dev)
    self.stream.wait_stream(cur)            # inputs produced on the caller's stream
    with torch.cuda.stream(self.stream):
      if timing:
        ev[0].record()
      pred = self.net.forward(images, use_graph=self.use_graph,
                              sparse=self.sparse_heads)
      if timing and not self.sparse_heads:
        ev[1].record()
      if S:
        if S > self.max_slots:
          raise ValueError('too many slots (%d > %d)' % (S, self.max_slots))
        Ksl = np.asarray(Ks, np.float64).reshape(B, 9)[[s[0] for s in slots]]
        sd = [(seed * 1000003 + (image_ids[im] if image_ids is not None else im)
               * 1009 + obj) & 0x7fffffffffffffff for im, obj in slots]
        ml = self._meta_layout
        self._view(mh, ml, 'Ks')[:S * 9] = torch.from_numpy(Ksl.reshape(-1))
        self._view(mh, ml, 'seeds')[:S] = torch.tensor(sd, dtype=torch.int64)
        self._view(mh, ml, 'slots')[:S * 2] = torch.tensor(
            slots, dtype=torch.int32).reshape(-1)
        self._view(mh, ml, 'max_models')[:S] = torch.tensor(wants,
                                                            dtype=torch.int32)
        self.meta_dev.copy_(mh, non_blocking=True)          # one H2D
        self.corr.S = S
        if self.sparse_heads:
          self.head_flops = self.net.run_sparse_heads(slots, self.corr.slots)
          if timing:
            ev[1].record()
        if after_net is not None:
          after_net(self)
        self.corr.count(pred[W.PRED_OBJ_CONF], pred[W.PRED_FRAG_CONF],
                        self.tau_a, self.tau_b)
        self.corr.fill(pred[W.PRED_OBJ_CONF], pred[W.PRED_FRAG_CONF],
                       pred[W.PRED_FRAG_LOC], self.output_scale)
        if timing:
          ev[2].record()
        if self.fitting_method == 'opencv_ransac':
          # one pose per slot: poses [S, 1, 12], num_models = success flag, scores 0.0; the
          # inlier mask (u8 per correspondence) lands in the bytes of the label buffer
          self.scores.zero_()
          _lib.check(self.lib.epos_solve_pnp_ransac_device(
              _ptr(self.corr.coord_2d), _ptr(self.corr.coord_3d),
              _ptr(self.corr.slot_base), S, self.corr.capacity, _ptr(self.Ks),
              ctypes.byref(self.cv_params), _ptr(self.work), _ptr(self.poses),
              _ptr(self.num_models), _ptr(self.labels), None,
              ctypes.c_void_p(self.stream.cuda_stream)), 'epos_solve_pnp_ransac_device')
        else:
          _lib.check(self.lib.epos_find6d_poses_device(
              _ptr(self.corr.coord_2d), _ptr(self.corr.coord_3d),
              _ptr(self.corr.slot_base), S, self.corr.capacity, _ptr(self.Ks),
              _ptr(self.max_models), _ptr(self.seeds), ctypes.byref(self.fit),
              max_k, _ptr(self.work), _ptr(self.poses), _ptr(self.scores),
              _ptr(self.num_models), _ptr(self.labels),
              ctypes.c_void_p(self.stream.cuda_stream)), 'epos_find6d_poses_device')
        # NB: poses / scores are laid out [S, max_k(call), ...] for this call.
        if timing:
          ev[3].record()
        res_host.copy_(self.res_dev, non_blocking=True)       # one D2H
      done.record()
    self._launches += 1
    self._pending.append((slots, wants, max_k, image_ids, scene_ids, timing, q))

  def collect(self):
    """Waits for the batch enqueued by ``launch()``; returns (poses, run_times)
    like process_image (infer.py:348-554)."""
    if not self._pending:
      raise _lib.EposError('collect() without launch()')
    slots, wants, max_k, image_ids, scene_ids, timing, q = self._pending.popleft()
    self._dones[q].synchronize()              # the one synchronisation
    S = len(slots)
    self.last_cap_hits = []
    poses_out = []
    if S:
      rh, rl = self._res_hosts[q], self._res_layout
      if int(self._view(rh, rl, 'overflow')[0]):
        raise _lib.EposError(
            'correspondence capacity (%d rows) exceeded' % self.corr.capacity)
      nm = self._view(rh, rl, 'num_models')[:S].numpy()
      if (nm < 0).any():
        raise _lib.EposError(
            'fitting stage: a hand-off between cooperating workgroups timed out for slot(s) '
            '%s (num_models == -1, include/epos_hip.h); repeat the step' %
            np.nonzero(nm < 0)[0].tolist())
      ph = self._view(rh, rl, 'poses')[:S * max_k * 12].numpy().reshape(
          S, max_k, 12)
      sh = self._view(rh, rl, 'scores')[:S * max_k].numpy().reshape(S, max_k)
      self.last_totals = self._view(rh, rl, 'totals')[:S * 2].numpy().reshape(
          S, 2).copy()
      for s, (im, obj_id) in enumerate(slots):
        if wants[s] < 0 and int(nm[s]) >= max_k:
          # "all found" (detection) stopped at the plan's instance cap: recorded for EVERY
          # (frame, object) -- cap_hits accumulates over the run, last_cap_hits is this
          # batch's -- and warned about once
          hit = (scene_ids[im] if scene_ids is not None else 0,
                 image_ids[im] if image_ids is not None else im, obj_id, int(nm[s]))
          self.cap_hits.append(hit)
          self.cap_hit_count += 1
          self.last_cap_hits.append(hit)
          if not self._warned_cap:
            import warnings
            warnings.warn('object %d: %d instances found = the cap this pipeline was '
                          'built with (max_instances); more may exist' %
                          (obj_id, int(nm[s])))
            self._warned_cap = True
        for i in range(int(nm[s])):
          poses_out.append({
              'scene_id': scene_ids[im] if scene_ids is not None else 0,
              'im_id': image_ids[im] if image_ids is not None else im,
              'obj_id': obj_id,
              'R': ph[s, i, :9].reshape(3, 3).copy(),
              't': ph[s, i, 9:].reshape(3, 1).copy(),
              'score': float(sh[s, i]),
          })
    run_times = {}
    if timing and S:
      e = self._evs[q]
      run_times = {'prediction': e[0].elapsed_time(e[1]) * 1e-3,
                   'establish_corr': e[1].elapsed_time(e[2]) * 1e-3,
                   'fitting': e[2].elapsed_time(e[3]) * 1e-3}
      run_times['total'] = sum(run_times.values())
      for p in poses_out:
        p['time'] = run_times['total']
    return poses_out, run_times

  def process_batch(self, images, Ks, targets, task_type=LOCALIZATION,
                    image_ids=None, scene_ids=None, seed=0, timing=False, after_net=None):
    """images f32 [B,H,W,3]; Ks [B,3,3]; targets per image {obj_id: count}.
    Returns (poses, run_times) like process_image (infer.py:348-554)."""
    self.launch(images, Ks, targets, task_type, image_ids, scene_ids, seed,
                timing, after_net)
    return self.collect()
