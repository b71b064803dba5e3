"""EPOS network (DeepLabv3+/Xception-65) executor for MI355X.

Builds, once per (batch, height, width, num_objs, num_frags), a static plan of
fused HIP launches for the forward pass that the reference builds as a TF graph
in ``model.predict`` (model.py:629-687 -> multi_scale_logits :517 -> get_logits
:461 -> feature.extract_features -> net_xception.xception_65), allocates every
activation buffer in HBM up front (NHWC fp32, the reference's layout and dtype)
and replays the plan on the current HIP stream -- eagerly or as a captured
hipGraph (no tracing compiler: the plan is explicit).

Fusion groups:
  * BatchNorm is folded into the preceding conv at weight-load time;
  * depthwise 3x3 (+BN, ReLU before/after) is one launch;
  * pointwise 1x1 (+BN, +residual add, +ReLU) is one fp32-MFMA GEMM launch that
    can read/write channel slices of the ASPP / decoder concat buffers, so no
    concat copy exists;
  * dense 3x3 stem convs: stride 1 with Cin % 32 == 0 (conv1_2) is an implicit GEMM
    inside the LDS-DMA kernel; the 3-channel stride-2 conv1_1 is im2col + the same GEMM.
torch is used for device memory and streams only.
"""
import ctypes
import os

import numpy as np
import torch

from epos_amd import _lib
from epos_amd import weights as W

XCEPTION_BN_EPS = 1e-3   # feature.py:300-307
HEAD_BN_EPS = 1e-5       # model.py:194-199, 307-312
RESNET_BN_EPS = 1e-5     # feature.py:282-287


def _ptr(t, offset_elems=0):
  return ctypes.c_void_p(t.data_ptr() + offset_elems * t.element_size())


def scale_dimension(dim, scale):
  """model.py:100-114."""
  return int((float(dim) - 1.0) * scale + 1.0)


_CAPTURE_STREAMS = {}


def _capture_stream(dev):
  """One shared side stream per device for graph capture from the default stream."""
  key = str(dev)
  if key not in _CAPTURE_STREAMS:
    _CAPTURE_STREAMS[key] = torch.cuda.Stream(dev)
  return _CAPTURE_STREAMS[key]


class EposNet(object):
  """Static-shape forward plan. ``forward(images)`` returns the logits buffers;
  ``predict(images)`` the reference's prediction dict (model.py:629-687)."""

  MAX_SLOTS = 512

  def __init__(self, checkpoint, batch, height, width, num_objs, num_frags=64,
               model_variant='xception_65', encoder_output_stride=8,
               decoder_output_stride=4, atrous_rates=(12, 24, 36),
               multi_grid=None, device='cuda:0', dry_run=False):
    if model_variant not in ('xception_65', 'resnet_v1_101_beta'):
      raise ValueError('Unsupported model variant: %s' % model_variant)
    self.model_variant = model_variant
    if encoder_output_stride != 8 or decoder_output_stride != 4:
      raise ValueError('Only encoder OS 8 / decoder OS 4 (common.py:127-135).')
    # dry_run: build the plan's STRUCTURE only (tests/test_graph_trace.py) -- buffers are
    # shape-only 'meta' tensors, no weight is packed, nothing can be launched.
    self.dry_run = bool(dry_run)
    if not self.dry_run and not torch.cuda.is_available():
      raise _lib.EposError('EposNet needs a HIP device (no CPU fallback).')
    self.lib = None if self.dry_run else _lib.load()
    self.dev = torch.device('meta' if self.dry_run else device)
    self.B, self.H, self.W = batch, height, width
    self.num_objs, self.num_frags = num_objs, num_frags
    self.atrous_rates = tuple(atrous_rates)
    self.multi_grid = list(multi_grid) if multi_grid else [1, 1, 1]
    self.ckpt = checkpoint
    self._keep = []          # device tensors owned by the plan
    self.ops = []            # (name, callable(stream))
    self.flops = 0           # multiply-add * 2 of the whole plan
    self.op_flops = {}
    self.op_kind = {}        # 'gemm' | 'dw' | 'im2col' | 'other'
    self.op_bytes = {}       # GEMM launches: algorithmic bytes (A + W + out + residual)
    # fusion-group byte model (algorithmic_bytes): per GEMM (A bytes, the rest, id of the A
    # buffer); per depthwise output buffer the bytes its launch READS (input + weights)
    self.op_io = {}
    self._dw_reads = {}
    self._graph = None
    self._graph_sparse = None
    self._graph_alt, self.alt_skip = None, None   # measurement aid: see capture_alt()
    # Absmax slots (include/epos_hip.h): the fp16-pair GEMM scales its fp32 A operand by a
    # power of two taken from an upper bound of max|A|; the producers of every activation
    # tensor keep that bound in a slot (GEMM epilogues by atomic max). `_bounds` maps a
    # buffer to (slot, slot2, gain, bias): bound = gain * max(slot, slot2) + bias. The
    # table is zeroed by the plan's first op.
    self._amax_table = torch.zeros(self.MAX_SLOTS * _lib.AMAX_WORDS, dtype=torch.int32,
                                   device=self.dev)
    self._n_slots = 0
    self._bounds = {}
    self.h2_layers, self.h2_refused = [], []
    # fp16-pair GEMM switches of the library (A/B runs): with either off no layer gets
    # fp16-pair weights. EPOS_H2_PRESPLIT (default 1 since round 4, 0 = off): the depthwise
    # kernels write their outputs already split (fp16 pairs), so each activation is converted
    # once instead of once per column tile of the GEMM and the GEMM loop carries no
    # conversion. Bit-identical results. Per launch the GEMMs gain 7 % (34.9 vs 37.4 us on
    # average over the plan, profiles/r04/presplit_ab.txt) while the depthwise launches pay
    # most of it back inside the pipelined step (+0.055 ms of depthwise slot time vs -0.02 ms
    # of GEMM and -0.05 ms of the rest): 417.5 / 419.8 vs 415.4 / 417.2 images/s, same box --
    # a small but repeatable gain (round 3 measured it neutral and kept it off).
    self.use_h2 = (os.environ.get('EPOS_GEMM_H2', '1') != '0' and
                   os.environ.get('EPOS_GEMM_SPLIT', '1') != '0')
    self.use_presplit = self.use_h2 and os.environ.get('EPOS_H2_PRESPLIT', '1') == '1'
    self._dw_h2 = {}           # id(depthwise output) -> its (mutable) launch arguments
    self.presplit_layers = []
    # Structure trace: one record per parametrised layer and a canonical expression per
    # buffer (channel slices of concat buffers separately), in the grammar of
    # tests/golden/tf_recorder.py -- what each launch computes, written down from the very
    # arguments the launch is built from. tests/test_graph_trace.py holds it against the
    # graph the reference's own code builds (tests/golden/graph_*.json).
    self.trace_layers, self.trace_outputs = [], {}
    self._exprs = {}
    self._last_bn = (None, None)
    self.pad_rows = os.environ.get('EPOS_PAD_ROWS', '1') != '0'
    self._lds = {}
    self._build_plan()

  # ------------------------------------------------------------ buffers ---
  def _empty(self, *shape, dtype=torch.float32):
    t = torch.empty(*shape, dtype=dtype, device=self.dev)
    self._keep.append(t)
    return t

  def _act(self, b, h, w, c):
    """Activation buffer [b, h, w, ld] for c channels: rows padded to a multiple of 32 floats
    (128-byte lines) where c is not one already -- Xception's 728-channel tensors become 736
    wide (+1.1 %) -- so that every pixel's channel vector starts on a line: the depthwise
    kernel's per-XCD channel slices then share no line (csrc/layers.hip; a dense 728-float
    row has 7 slice boundaries inside lines and ~30 % of the input is fetched by two XCDs),
    and the GEMMs' A rows and C rows are line-aligned. The padding columns are never read or
    written. EPOS_PAD_ROWS=0: dense rows (the A/B switch)."""
    ld = (c + 31) // 32 * 32 if (self.pad_rows and c >= 256) else c
    t = self._empty(b, h, w, ld)
    self._lds[id(t)] = ld
    return t

  def _ld(self, buf):
    """Row pitch (floats) of an activation buffer."""
    return self._lds.get(id(buf), buf.shape[-1])

  def _dev(self, arr):
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.dev)
    self._keep.append(t)
    return t

  # ---------------------------------------------------- structure trace ---
  def _set_expr(self, buf, expr, off=0, width=None):
    width = buf.shape[-1] if width is None else width
    ent = [e for e in self._exprs.get(id(buf), []) if e[0] != off]
    ent.append((off, width, expr))
    self._exprs[id(buf)] = sorted(ent)

  def _expr_of(self, buf, off=0, width=None):
    width = buf.shape[-1] if width is None else width
    ent = self._exprs[id(buf)]
    for o, w, e in ent:
      if o == off and w == width:
        return e
    parts, at = [], off
    for o, w, e in ent:                   # a read across the slices of a concat buffer
      if o == at and at < off + width:
        parts.append(e)
        at += w
    assert at == off + width and parts, ('no expression for the slice', off, width, ent)
    return 'concat(%s)' % ','.join(parts)

  @staticmethod
  def _relu_expr(e):
    return e if e.startswith('relu(') else 'relu(%s)' % e

  def _trace_layer(self, scope, op, kernel, stride, rate, padding, cin, cout, bn_eps,
                   bias, expr_in, out_hw):
    self.trace_layers.append({
        'scope': scope, 'op': op, 'kernel': [kernel, kernel], 'stride': stride,
        'rate': rate, 'padding': padding, 'cin': cin, 'cout': cout, 'bn_eps': bn_eps,
        'bias': bias, 'input': expr_in, 'out_hw': [int(out_hw[0]), int(out_hw[1])]})

  # ------------------------------------------------------- absmax slots ---
  def _new_slot(self):
    i = self._n_slots
    if i >= self.MAX_SLOTS:
      raise _lib.EposError('absmax slot table exhausted')
    self._n_slots += 1
    return i

  def _slot_ptr(self, i):
    return _ptr(self._amax_table, i * _lib.AMAX_WORDS) if i is not None else None

  def _bound_of(self, buf):
    """(slot, slot2, gain, bias) of a buffer, or None when nobody tracks it."""
    return self._bounds.get(id(buf))

  def _set_bound(self, buf, slot, slot2=None, gain=0.0, bias=0.0):
    self._bounds[id(buf)] = (slot, slot2, float(gain), float(bias))

  def _out_slot(self, buf, n, ldc, off=0):
    """Slot that a GEMM writing `buf` publishes max|out| into, or None when its epilogue
    cannot (rows not float4-able). Writers of one (concat) buffer share the slot."""
    if n % 4 or ldc % 4 or off % 4:
      return None
    b = self._bound_of(buf)
    if b is not None and b[1] is None and b[2] == 0.0:
      return b[0]
    slot = self._new_slot()
    self._set_bound(buf, slot)
    return slot

  # ------------------------------------------------------ weight packing ---
  def _pack_pointwise(self, w_kn, scale, bias):
    """w_kn [K, N] (TF HWIO with H=W=1), BN scale folded, packed for the GEMM."""
    k, n = w_kn.shape
    if self.dry_run:
      return self._empty(1), self._empty(1), (k + 3) // 4 * 4
    w = np.ascontiguousarray(w_kn.astype(np.float32) * scale[None, :].astype(
        np.float32))
    kpad = (k + 3) // 4 * 4
    if kpad != k:
      w = np.concatenate([w, np.zeros((kpad - k, n), np.float32)], 0)
    total = self.lib.epos_pack_pointwise_weights(None, kpad, n, None)
    dst = np.empty(total, np.float32)
    self.lib.epos_pack_pointwise_weights(
        w.ctypes.data_as(ctypes.c_void_p), kpad, n,
        dst.ctypes.data_as(ctypes.c_void_p))
    npad = (n + 127) // 128 * 128
    b = np.zeros(npad, np.float32)
    b[:n] = bias
    return self._dev(dst), self._dev(b), kpad

  def _pack_split(self, w_kn, scale):
    """The same folded weights in the split-operand GEMM's layout (three exact bf16
    pieces per weight, MFMA fragment order): epos_pack_pointwise_weights_split."""
    if self.dry_run:
      return self._empty(1)
    k, n = w_kn.shape
    w = np.ascontiguousarray(w_kn.astype(np.float32) * scale[None, :].astype(
        np.float32))
    kpad = (k + 3) // 4 * 4
    if kpad != k:
      w = np.concatenate([w, np.zeros((kpad - k, n), np.float32)], 0)
    total = self.lib.epos_pack_pointwise_weights_split(None, kpad, n, None)
    dst = np.empty(total, np.uint8)
    self.lib.epos_pack_pointwise_weights_split(
        w.ctypes.data_as(ctypes.c_void_p), kpad, n,
        dst.ctypes.data_as(ctypes.c_void_p))
    return self._dev(dst)

  def _pack_h2(self, w_kn, scale):
    """The same folded weights as fp16 pairs with per-column power-of-two scales
    (epos_pack_pointwise_weights_h2), or None when the matrix is refused there (a weight
    outside the window fp16 pairs represent to 2^-22): the layer then stays on the
    bf16 x 6 kernel."""
    if self.dry_run:
      return self._empty(1)
    k, n = w_kn.shape
    w = np.ascontiguousarray(w_kn.astype(np.float32) * scale[None, :].astype(
        np.float32))
    kpad = (k + 3) // 4 * 4
    if kpad != k:
      w = np.concatenate([w, np.zeros((kpad - k, n), np.float32)], 0)
    total = self.lib.epos_pack_pointwise_weights_h2(
        w.ctypes.data_as(ctypes.c_void_p), kpad, n, None)
    if total <= 0:
      return None
    dst = np.empty(total, np.uint8)
    self.lib.epos_pack_pointwise_weights_h2(
        w.ctypes.data_as(ctypes.c_void_p), kpad, n,
        dst.ctypes.data_as(ctypes.c_void_p))
    return self._dev(dst)

  def _conv_params(self, scope, eps):
    """1x1 / dense conv followed by BN -> (w [K,N], scale, bias)."""
    w = self.ckpt[scope + '/weights']
    kh, kw, cin, cout = w.shape
    scale, bias = W.fold_bn(self.ckpt, scope, eps, 'conv')
    self._last_bn = (scope, eps)
    return w.reshape(kh * kw * cin, cout), scale, bias

  def _dw_params(self, scope, eps):
    w = self.ckpt[scope + '/depthwise_weights']          # [3,3,C,1]
    scale, bias = W.fold_bn(self.ckpt, scope, eps, 'dw')
    w9c = (w[:, :, :, 0].reshape(9, -1) * scale[None, :]).astype(np.float32)
    # |depthwise output| <= gain * max|input| + bias0 (the consumer GEMM's A bound)
    self._dw_gain = float(np.abs(w9c.astype(np.float64)).sum(0).max())
    self._dw_bias0 = float(np.abs(np.asarray(bias, np.float64)).max())
    self._last_bn = (scope, eps)
    return self._dev(w9c), self._dev(bias)

  # --------------------------------------------------------------- ops ---
  def _add(self, name, fn, flops=0, kind='other', nbytes=0):
    self.op_bytes[name] = nbytes
    self.ops.append((name, fn))
    self.flops += flops
    self.op_flops[name] = flops
    self.op_kind[name] = kind

  def _pointwise(self, name, a, a_off, lda, m, k, w_kn, scale, bias, c, c_off,
                 ldc, relu, relu_in=False, res=None, res_off=0, ldr=0, sub=1,
                 ho=0, wo=0, hi=0, wi=0, group=None, track_out=True,
                 trace=True):
    """One 1x1 conv. With ``group`` (a list) the problem is only appended to it;
    ``_flush_group`` later launches the whole list as ONE grouped GEMM."""
    wp, bp, kpad = self._pack_pointwise(w_kn, scale, bias)
    ws = None if relu_in else self._pack_split(w_kn, scale)
    n = w_kn.shape[1]
    assert kpad == k or (kpad > k and lda >= kpad), (name, k, kpad, lda)
    # ---- structure trace: what this launch computes, from its own arguments. A stem conv
    # that runs as im2col + GEMM is recorded by _stem_conv (it passes trace=False).
    if trace:
      ein = self._expr_of(a, a_off, k)
      if relu_in:
        ein = self._relu_expr(ein)
      bn_eps = self._last_bn[1] if self._last_bn[0] == name else None
      hw = (ho, wo) if sub > 1 else (c.shape[1:3] if c.dim() == 4 else (1, 1))
      self._trace_layer(name, 'conv2d', 1, sub, 1, 'SAME', k, n, bn_eps, bn_eps is None,
                        ein, hw)
      eout = 'L:' + name
      if res is not None:
        eout = 'add(%s)' % ','.join(sorted([eout, self._expr_of(res, res_off, n)]))
      self._set_expr(c, self._relu_expr(eout) if relu else eout, c_off, n)
    # fp16-pair weights when the A operand has a bound; the output's slot
    ab = self._bound_of(a) if self.use_h2 else None
    wh = None
    if ab is not None and not relu_in and m > 8:
      wh = self._pack_h2(w_kn, scale)
      (self.h2_layers if wh is not None else self.h2_refused).append(name)
    # A = the output of a depthwise conv that was set up to write fp16 pairs: keep that
    # only if this GEMM really runs on the fp16-pair kernel (the packer may have refused
    # the weights); the depthwise arguments are the very struct its launch closure holds
    dwa = self._dw_h2.pop(id(a), None)
    presplit = False
    if dwa is not None:
      presplit = wh is not None and sub == 1 and a_off == 0
      dwa.y_h2 = int(presplit)
      if presplit:
        self.presplit_layers.append(name)
    track = track_out and not relu_in and m > 8 and (res is None or ldr % 4 == 0)
    cslot = self._out_slot(c, n, ldc, c_off) if track else None
    args = _lib.PointwiseArgs(
        A=_ptr(a, a_off), lda=lda, Wp=_ptr(wp), bias=_ptr(bp),
        R=_ptr(res, res_off) if res is not None else None, ldr=ldr,
        C=_ptr(c, c_off), ldc=ldc, M=m, N=n, K=kpad, relu=int(relu),
        relu_in=int(relu_in), sub=sub, Ho=ho, Wo=wo, Hi=hi, Wi=wi,
        Ws=_ptr(ws) if ws is not None else None,
        Wh=_ptr(wh) if wh is not None else None,
        a_amax=self._slot_ptr(ab[0]) if wh is not None else None,
        a_amax2=self._slot_ptr(ab[1]) if wh is not None else None,
        a_gain=ab[2] if wh is not None else 0.0,
        a_bias=ab[3] if wh is not None else 0.0,
        a_presplit=int(presplit),
        c_amax=self._slot_ptr(cslot))
    lib = self.lib
    # fp32 activations in and out, weights once (4 B each: what the layer IS; the
    # split kernel streams 6 B per weight), residual once
    nbytes = 4 * (m * k + k * n + m * n + (m * n if res is not None else 0))
    self.op_io[name] = (4 * m * k, nbytes - 4 * m * k, id(a))
    if group is not None:
      group.append((name, args, 2 * m * n * k, nbytes))
      return

    def run(stream, args=args):
      _lib.check(lib.epos_pointwise_conv_grouped_f32(ctypes.byref(args), 1, stream), name)
    self._add(name, run, 2 * m * n * k, 'gemm', nbytes)
    # the launch closure holds `args` itself: _build_plan may still attach the image-pooling
    # block sums to the launch that writes the encoder output
    self._last_pw = (c, args, wh is not None and res is None and c_off == 0 and n % 4 == 0)

  def _flush_group(self, group):
    """Launches the collected problems as one grouped GEMM (they must agree on
    relu_in / residual; callers group accordingly)."""
    if not group:
      return
    # problems whose A is already fp16 pairs run on another kernel instantiation than the
    # ones that split their fp32 A themselves: one launch per kind
    kinds = sorted({int(g[1].a_presplit) for g in group})
    if len(kinds) > 1:
      for kd in kinds:
        self._flush_group([g for g in group if int(g[1].a_presplit) == kd])
      del group[:]
      return
    name = '+'.join(g[0] for g in group)
    arr = (_lib.PointwiseArgs * len(group))(*[g[1] for g in group])
    flops = sum(g[2] for g in group)
    nbytes = sum(g[3] for g in group)
    lib = self.lib
    n = len(group)

    def run(stream, arr=arr):
      _lib.check(lib.epos_pointwise_conv_grouped_f32(arr, n, stream), name)
    self._add(name, run, flops, 'gemm', nbytes)
    del group[:]

  def _depthwise(self, name, x, ldx, hi, wi, c, stride, rate, scope, eps,
                 relu_in, relu_out):
    """One depthwise 3x3 launch."""
    ho = hi if stride == 1 else (hi - 1) // 2 + 1
    wo = wi if stride == 1 else (wi - 1) // 2 + 1
    w9c, bias = self._dw_params(scope, eps)
    y = self._act(self.B, ho, wo, c)
    ldy = self._ld(y)
    ein = self._expr_of(x, 0, c)
    if relu_in:
      ein = self._relu_expr(ein)
    if stride > 1:                # fixed_padding (net_xception.py:74-93) + VALID
      ein = 'pad(%s,%d,%d)' % (ein, rate, rate)
    self._trace_layer(name, 'depthwise_conv2d', 3, stride, rate,
                      'SAME' if stride == 1 else 'VALID', c, c, eps, False, ein, (ho, wo))
    self._set_expr(y, self._relu_expr('L:' + name) if relu_out else 'L:' + name, 0, c)
    xb = self._bound_of(x)
    if xb is not None:
      g, b0 = self._dw_gain, self._dw_bias0
      self._set_bound(y, xb[0], xb[1], g * (xb[2] if xb[2] else 1.0),
                      g * xb[3] + b0)
    args = _lib.DepthwiseArgs(
        X=_ptr(x), ldx=ldx, w9c=_ptr(w9c), bias=_ptr(bias), Y=_ptr(y), ldy=ldy,
        B=self.B, Hi=hi, Wi=wi, Ho=ho, Wo=wo, C=c, stride=stride, rate=rate,
        relu_in=int(relu_in), relu_out=int(relu_out))
    yb = self._bound_of(y)
    if yb is not None and self.use_presplit:
      # fp16-pair output, pending the consumer's decision (_pointwise): scale from the
      # bound of |Y| = gain * max|X| + max|bias| (the same numbers the GEMM gets)
      args.y_h2 = 1
      args.x_amax, args.x_amax2 = self._slot_ptr(yb[0]), self._slot_ptr(yb[1])
      args.gain, args.bias0 = yb[2], yb[3]
      self._dw_h2[id(y)] = args
    lib = self.lib
    self._dw_reads[id(y)] = 4 * (self.B * hi * wi * c + 10 * c)
    def run(stream, args=args):
      _lib.check(lib.epos_depthwise3x3_f32(ctypes.byref(args), stream), name)
    # algorithmic bytes: the input read once + the output written once (fp32), weights
    self._add(name, run, 2 * 9 * self.B * ho * wo * c, 'dw',
              4 * (self.B * hi * wi * c + self.B * ho * wo * c + 10 * c))
    return y, ho, wo

  def _stem_conv(self, name, x, hi, wi, cin, scope, stride, preprocess,
                 eps=XCEPTION_BN_EPS, rate=1):
    """resnet_utils.conv2d_same 3x3 (+BN+ReLU) = im2col + GEMM
    (net_xception.py:460-463; net_resnet_v1_beta.py:82-83,108-110). stride 1 ->
    'SAME' (pad = rate); stride 2 -> fixed_padding + VALID (pad = rate)."""
    ho = hi if stride == 1 else (hi - 1) // 2 + 1
    wo = wi if stride == 1 else (wi - 1) // 2 + 1
    k = 9 * cin
    ein = self._expr_of(x, 0, cin)
    if preprocess:                # fused into the im2col (feature.py:171-174)
      assert ein == 'input'
      ein = 'preprocess(input)'
    if stride > 1:                # conv2d_same: explicit padding + VALID
      ein = 'pad(%s,%d,%d)' % (ein, rate, rate)
    cout_t = self.ckpt[scope + '/weights'].shape[3]
    self._trace_layer(name, 'conv2d', 3, stride, rate, 'SAME' if stride == 1 else 'VALID',
                      cin, cout_t, eps, False, ein, (ho, wo))
    if cin % 32 == 0 and not preprocess:
      # implicit GEMM: the LDS-DMA kernel gathers the shifted input pixels itself
      w_kn, scale, bias = self._conv_params(scope, eps)
      wp, bp, kpad = self._pack_pointwise(w_kn, scale, bias)
      ws = self._pack_split(w_kn, scale)
      cout = w_kn.shape[1]
      y = self._empty(self.B, ho, wo, cout)
      xb = self._bound_of(x)
      wh = None
      if xb is not None and xb[1] is None and xb[2] == 0.0:   # a plain slot
        wh = self._pack_h2(w_kn, scale)
        (self.h2_layers if wh is not None else self.h2_refused).append(name)
      yslot = self._out_slot(y, cout, cout)
      cargs = _lib.Conv3x3Args(X=_ptr(x), ldx=cin, Wp=_ptr(wp), bias=_ptr(bp),
                               Y=_ptr(y), ldy=cout, B=self.B, H=hi, W=wi, Cin=cin,
                               Cout=cout, stride=stride, rate=rate, relu=1,
                               Ws=_ptr(ws), Wh=_ptr(wh) if wh is not None else None,
                               x_amax=self._slot_ptr(xb[0]) if wh is not None else None,
                               y_amax=self._slot_ptr(yslot))
      lib = self.lib

      def run_conv(stream, cargs=cargs):
        _lib.check(lib.epos_conv3x3_f32(ctypes.byref(cargs), stream), name)
      self._add(name, run_conv, 2 * self.B * ho * wo * cout * k, 'gemm',
                4 * (self.B * hi * wi * cin + k * cout + self.B * ho * wo * cout))
      self.op_io[name] = (4 * self.B * hi * wi * cin,
                          4 * (k * cout + self.B * ho * wo * cout), id(x))
      self._set_expr(y, 'relu(L:%s)' % name)
      return y, ho, wo, cout
    ldcol = (k + 3) // 4 * 4
    m = self.B * ho * wo
    col = self._empty(m, ldcol)
    args = _lib.Im2colArgs(
        X=_ptr(x), ldx=cin, col=_ptr(col), ldcol=ldcol, B=self.B, Hi=hi, Wi=wi,
        Ho=ho, Wo=wo, C=cin, stride=stride, rate=rate, pad=rate,
        preprocess=int(preprocess))
    lib = self.lib
    if getattr(self, '_first_im2col', None) is None:
      self._first_im2col = (name + '/im2col', args)

    def run(stream, args=args):
      _lib.check(lib.epos_im2col3x3_f32(ctypes.byref(args), stream), name)
    self._add(name + '/im2col', run, 0, 'im2col')
    w_kn, scale, bias = self._conv_params(scope, eps)
    cout = w_kn.shape[1]
    y = self._empty(self.B, ho, wo, cout)
    self._pointwise(name, col, 0, ldcol, m, k, w_kn, scale, bias, y, 0, cout,
                    relu=True, trace=False)
    self._set_expr(y, 'relu(L:%s)' % name)
    return y, ho, wo, cout

  # ------------------------------------------------ ResNet-v1-101-beta (C5) ---
  def _simple(self, name, fn):
    self._add(name, fn)

  def _bottleneck(self, scope, x, hi, wi, cin, depth, db, stride, rate,
                  keep_conv3=False):
    """net_resnet_v1_beta.py:38-93: 1x1 -> 3x3 (conv2d_same, rate) -> 1x1, plus
    shortcut, ReLU after the add. The add + ReLU is the epilogue of the conv3 GEMM
    unless conv3 itself is an end point (decoder tap, feature.py:50-54)."""
    eps = RESNET_BN_EPS
    B, lib = self.B, self.lib
    ho = hi if stride == 1 else (hi - 1) // 2 + 1
    wo = wi if stride == 1 else (wi - 1) // 2 + 1
    m_in, m_out = B * hi * wi, B * ho * wo
    if depth == cin:
      if stride == 1:
        shortcut = x
      else:                                    # resnet_utils.subsample (:71-72)
        shortcut = self._empty(B, ho, wo, depth)

        def run_sub(stream, x=x, y=shortcut):
          _lib.check(lib.epos_subsample_f32(_ptr(x), cin, _ptr(y), depth, B, hi,
                                            wi, cin, stride, stream), 'subsample')
        self._add(scope + '/shortcut_subsample', run_sub)
        self._glue_bytes = getattr(self, '_glue_bytes', 0) + 8 * B * ho * wo * depth
        self._set_expr(shortcut, 'subsample(%s,%d)' % (self._expr_of(x, 0, cin), stride))
        if self._bound_of(x) is not None:
          self._set_bound(shortcut, *self._bound_of(x))
    else:
      w_kn, sc, bi = self._conv_params(scope + '/shortcut', eps)
      shortcut = self._empty(B, ho, wo, depth)
      self._pointwise(scope + '/shortcut', x, 0, cin, m_out, cin, w_kn, sc, bi,
                      shortcut, 0, depth, relu=False, sub=stride, ho=ho, wo=wo,
                      hi=hi, wi=wi)
    w_kn, sc, bi = self._conv_params(scope + '/conv1', eps)
    r1 = self._empty(B, hi, wi, db)
    self._pointwise(scope + '/conv1', x, 0, cin, m_in, cin, w_kn, sc, bi, r1, 0,
                    db, relu=True)
    r2, _, _, _ = self._stem_conv(scope + '/conv2', r1, hi, wi, db,
                                  scope + '/conv2', stride, False, eps=eps,
                                  rate=rate)
    w_kn, sc, bi = self._conv_params(scope + '/conv3', eps)
    out = self._empty(B, ho, wo, depth)
    conv3 = None
    if keep_conv3:
      conv3 = self._empty(B, ho, wo, depth)
      self._pointwise(scope + '/conv3', r2, 0, db, m_out, db, w_kn, sc, bi, conv3,
                      0, depth, relu=False)

      def run_add(stream, a=conv3, b=shortcut, y=out):
        _lib.check(lib.epos_add_relu_f32(_ptr(a), _ptr(b), _ptr(y),
                                         m_out * depth, stream), 'add_relu')
      self._add(scope + '/add_relu', run_add)
      self._glue_bytes = getattr(self, '_glue_bytes', 0) + 12 * m_out * depth
      self._set_expr(out, 'relu(add(%s))' % ','.join(sorted(
          [self._expr_of(conv3), self._expr_of(shortcut)])))
      oslot = self._new_slot()
      self._set_bound(out, oslot)

      def run_amax(stream, y=out, oslot=oslot):
        _lib.check(lib.epos_absmax_f32(_ptr(y), depth, m_out, depth,
                                       self._slot_ptr(oslot), stream), 'add_relu/absmax')
      self._add(scope + '/add_relu/absmax', run_amax)
    else:
      self._pointwise(scope + '/conv3', r2, 0, db, m_out, db, w_kn, sc, bi, out,
                      0, depth, relu=True, res=shortcut, ldr=depth)
    return out, ho, wo, depth, conv3

  def _backbone_resnet(self):
    """resnet_v1_101_beta (net_resnet_v1_beta.py:445-516) at output_stride 8."""
    B, H, Wd, lib = self.B, self.H, self.W, self.lib
    net = 'resnet_v1_101'
    x, h, w, c = self.images, H, Wd, 3
    for i, stride in enumerate([2, 1, 1], 1):                # :108-110
      x, h, w, c = self._stem_conv('%s/conv1_%d' % (net, i), x, h, w, c,
                                   '%s/conv1_%d' % (net, i), stride, i == 1,
                                   eps=RESNET_BN_EPS)
    ph, pw = (h + 1) // 2, (w + 1) // 2
    pooled = self._empty(B, ph, pw, c)

    def run_pool(stream, x=x, y=pooled, h=h, w=w, c=c):
      _lib.check(lib.epos_maxpool3x3_s2_f32(_ptr(x), c, _ptr(y), c, B, h, w, c,
                                            stream), 'maxpool')
    self._add(net + '/pool1', run_pool)                      # :190
    self._glue_bytes = getattr(self, '_glue_bytes', 0) + 4 * B * (h * w + ph * pw) * c
    self._set_expr(pooled, 'maxpool(%s,3,2,SAME)' % self._expr_of(x, 0, c))
    if self._bound_of(x) is not None:          # a max-pool output is bounded by its input
      self._set_bound(pooled, *self._bound_of(x))
    x, h, w = pooled, ph, pw
    target, current_stride, rate = 2, 1, 1                   # 8 / 4 (:185-188)
    low_level = None
    mg = self.multi_grid
    for bscope, base, units in W.RESNET101_BLOCKS:
      for u in range(units):
        scope = '%s/%s/unit_%d/bottleneck_v1' % (net, bscope, u + 1)
        stride = 2 if (u == units - 1 and bscope != 'block4') else 1
        unit_rate = mg[u] if bscope == 'block4' else 1
        keep = bscope == 'block1' and u == 1
        if current_stride == target:
          x, h, w, c, conv3 = self._bottleneck(scope, x, h, w, c, base * 4, base,
                                               1, rate * unit_rate, keep)
          rate *= stride
        else:
          x, h, w, c, conv3 = self._bottleneck(scope, x, h, w, c, base * 4, base,
                                               stride, unit_rate, keep)
          current_stride *= stride
        if keep:
          low_level = (conv3, h, w, base * 4)
    return x, h, w, c, low_level

  def _backbone_xception(self):
    B, H, Wd = self.B, self.H, self.W
    net = 'xception_65'
    x, h, w, c = self._stem_conv(net + '/entry_flow/conv1_1', self.images, H, Wd,
                                 3, net + '/entry_flow/conv1_1', 2, True)
    x, h, w, c = self._stem_conv(net + '/entry_flow/conv1_2', x, h, w, c,
                                 net + '/entry_flow/conv1_2', 1, False)
    # stack_blocks_dense (net_xception.py:326-393) with output_stride 8/2 = 4.
    target, current_stride, rate = 4, 1, 1
    low_level = None
    blocks = [
        ('entry_flow/block1', [128, 128, 128], 'conv', False, 1, 2, [1, 1, 1]),
        ('entry_flow/block2', [256, 256, 256], 'conv', False, 1, 2, [1, 1, 1]),
        ('entry_flow/block3', [728, 728, 728], 'conv', False, 1, 2, [1, 1, 1]),
        ('middle_flow/block1', [728, 728, 728], 'sum', False, 16, 1, [1, 1, 1]),
        ('exit_flow/block1', [728, 1024, 1024], 'conv', False, 1, 2, [1, 1, 1]),
        ('exit_flow/block2', [1536, 1536, 2048], 'none', True, 1, 1,
         self.multi_grid),
    ]
    for bscope, depths, skip, act, units, stride, url in blocks:
      for u in range(units):
        scope = '%s/%s/unit_%d/xception_module' % (net, bscope, u + 1)
        # the decoder taps sep-conv 2 of entry_flow/block2 BEFORE any activation
        linear = (1,) if bscope == 'entry_flow/block2' else ()
        if current_stride == target:
          x, h, w, c, taps = self._xception_module(
              scope, x, h, w, c, depths, skip, act, 1, rate, url, linear)
          rate *= stride
        else:
          x, h, w, c, taps = self._xception_module(
              scope, x, h, w, c, depths, skip, act, stride, 1, url, linear)
          current_stride *= stride
        if bscope == 'entry_flow/block2':
          low_level = (taps[1], taps[1].shape[1], taps[1].shape[2], depths[1])
    return x, h, w, c, low_level

  def _xception_module(self, scope, x, hi, wi, cin, depths, skip, act_in_sep,
                       stride, rate, unit_rates, linear_taps=()):
    """net_xception.py:197-323 as 3 x (depthwise launch + GEMM launch); the skip
    connection is the residual input of the third GEMM's epilogue.

    In the pre-activation form (ReLU -> depthwise -> pointwise, :272-276) the outputs
    of sep-convs 1 and 2 are read by the next ReLU only, so that ReLU is applied by
    the producing GEMM's epilogue (same values, bit for bit) instead of on every
    tap the depthwise kernel loads -- except for `linear_taps`, which the decoder
    reads before the activation (feature.py:29-73 'entry_flow/block2/unit_1/
    xception_module/separable_conv2_pointwise')."""
    eps = XCEPTION_BN_EPS
    ho = hi if stride == 1 else (hi - 1) // 2 + 1
    wo = wi if stride == 1 else (wi - 1) // 2 + 1
    shortcut = None
    grp = []
    if skip == 'conv':
      # The shortcut GEMM shares a launch with the first pointwise conv.
      w_kn, sc, bi = self._conv_params(scope + '/shortcut', eps)
      shortcut = self._act(self.B, ho, wo, depths[2])
      self._pointwise(scope + '/shortcut', x, 0, self._ld(x), self.B * ho * wo, cin,
                      w_kn, sc, bi, shortcut, 0, self._ld(shortcut), relu=False,
                      sub=stride, ho=ho, wo=wo, hi=hi, wi=wi, group=grp)
    r, rh, rw, rc = x, hi, wi, cin
    taps = {}
    r_is_relu = False           # r already holds ReLU(previous sep-conv output)
    for i in range(3):
      sc = '%s/separable_conv%d' % (scope, i + 1)
      s_i = stride if i == 2 else 1
      d, dh, dw_ = self._depthwise(
          sc + '_depthwise', r, self._ld(r), rh, rw, rc, s_i, rate * unit_rates[i],
          sc + '_depthwise', eps, relu_in=(not act_in_sep) and not r_is_relu,
          relu_out=act_in_sep)
      w_kn, scl, bi = self._conv_params(sc + '_pointwise', eps)
      y = self._act(self.B, dh, dw_, depths[i])
      res, ldr = None, 0
      if i == 2 and skip == 'conv':
        res, ldr = shortcut, self._ld(shortcut)
      elif i == 2 and skip == 'sum':
        res, ldr = x, self._ld(x)
      fold_next_relu = (not act_in_sep) and i < 2 and i not in linear_taps
      self._pointwise(sc + '_pointwise', d, 0, self._ld(d), self.B * dh * dw_, rc, w_kn,
                      scl, bi, y, 0, self._ld(y), relu=act_in_sep or fold_next_relu,
                      res=res, ldr=ldr, group=grp if i == 0 else None)
      r_is_relu = fold_next_relu
      if i == 0:
        self._flush_group(grp)
      taps[i] = y
      r, rh, rw, rc = y, dh, dw_, depths[i]
    return r, rh, rw, rc, taps

  # -------------------------------------------------------------- plan ---
  def _build_plan(self):
    B, H, Wd = self.B, self.H, self.W
    self.images = self._empty(B, H, Wd, 3)
    self._images_u8 = None          # device staging for uint8 frames (set_images)
    self._set_expr(self.images, 'input')
    lib0 = self.lib

    def run_clear(stream):
      _lib.check(lib0.epos_amax_clear(_ptr(self._amax_table), self._n_slots, stream),
                 'amax_clear')
    self._add('amax_clear', run_clear)
    if self.model_variant == 'xception_65':
      x, h, w, c, low_level = self._backbone_xception()
    else:
      x, h, w, c, low_level = self._backbone_resnet()
    self.encoder = x
    eh, ew, ec = h, w, c
    lib = self.lib

    # ---- ASPP (model.py:213-265): branches write slices of one 1280-wide buffer.
    nb = 2 + len(self.atrous_rates)
    cat = self._empty(B, eh, ew, 256 * nb)
    ldcat = 256 * nb
    m_enc = B * eh * ew
    pooled = self._empty(B, ec)

    # Image pooling (model.py:220). Round 4: when the encoder output is written by ONE
    # fp16-pair GEMM launch without residual (Xception: exit_flow/block2 separable_conv3),
    # that launch's epilogue also writes the column sums of every block of 32 rows and a small
    # kernel finishes the mean from 150 x 2048 floats instead of re-reading the 39 MB tensor
    # (EPOS_POOL_FOLD=0: the stand-alone reduction, as for ResNet, whose last launch carries
    # a residual, and for batches whose images are not a whole number of 32-row blocks).
    lp = getattr(self, '_last_pw', None)
    fold_pool = (not self.dry_run and os.environ.get('EPOS_POOL_FOLD', '1') == '1' and
                 lp is not None and lp[0] is x and lp[2] and
                 ((eh * ew) % 32 == 0 or B == 1))
    self.pool_folded = bool(fold_pool)
    if fold_pool:
      blocks = (eh * ew + 31) // 32
      part = self._empty(B * blocks, ec)
      lp[1].col_sums = _ptr(part)
      lp[1].col_ld = ec

      def run_pool(stream, part=part, pooled=pooled):
        _lib.check(lib.epos_global_avg_pool_partial_f32(_ptr(part), ec, _ptr(pooled), B,
                                                        blocks, ec, eh * ew, stream),
                   'avg_pool_partial')
    else:
      def run_pool(stream, x=x, pooled=pooled):
        _lib.check(lib.epos_global_avg_pool_f32(_ptr(x), ec, _ptr(pooled), B,
                                                eh * ew, ec, stream), 'avg_pool')
    self._add('image_pooling/mean', run_pool)
    self._set_expr(pooled, 'mean(%s)' % self._expr_of(x, 0, ec))
    w_kn, sc, bi = self._conv_params('image_pooling', HEAD_BN_EPS)
    pool_feat = self._empty(B, 256)
    self._pointwise('image_pooling', pooled, 0, ec, B, ec, w_kn, sc, bi,
                    pool_feat, 0, 256, relu=True)

    def run_bcast(stream, pool_feat=pool_feat, cat=cat):
      _lib.check(lib.epos_resize_bilinear_f32(
          _ptr(pool_feat), 256, _ptr(cat), ldcat, B, 1, 1, eh, ew, 256, stream),
                 'image_pooling/resize')
    self._add('image_pooling/resize', run_bcast)
    self._set_expr(cat, 'resize(%s,%dx%d)' % (self._expr_of(pool_feat), eh, ew), 0, 256)
    # The four spatial ASPP branches (N = 256 each) share ONE grouped GEMM launch.
    grp = []
    w_kn, sc, bi = self._conv_params('aspp0', HEAD_BN_EPS)
    self._pointwise('aspp0', x, 0, ec, m_enc, ec, w_kn, sc, bi, cat, 256, ldcat,
                    relu=True, group=grp)
    for i, r in enumerate(self.atrous_rates, 1):
      d, _, _ = self._depthwise('aspp%d_depthwise' % i, x, ec, eh, ew, ec, 1, r,
                                'aspp%d_depthwise' % i, HEAD_BN_EPS, False, True)
      w_kn, sc, bi = self._conv_params('aspp%d_pointwise' % i, HEAD_BN_EPS)
      self._pointwise('aspp%d_pointwise' % i, d, 0, self._ld(d), m_enc, ec, w_kn, sc, bi,
                      cat, 256 * (i + 1), ldcat, relu=True, group=grp)
    self._flush_group(grp)
    # the broadcast image-pooling branch is part of `cat` too: its absmax joins the slot
    # the four GEMM problems publish into
    cb = self._bound_of(cat)
    if cb is not None:
      cslot = self._slot_ptr(cb[0])

      def run_pool_amax(stream, pool_feat=pool_feat):
        _lib.check(lib.epos_absmax_f32(_ptr(pool_feat), 256, B, 256, cslot, stream),
                   'image_pooling/absmax')
      self._add('image_pooling/absmax', run_pool_amax)
    w_kn, sc, bi = self._conv_params('concat_projection', HEAD_BN_EPS)
    proj = self._empty(B, eh, ew, 256)
    self._pointwise('concat_projection', cat, 0, ldcat, m_enc, ldcat, w_kn, sc,
                    bi, proj, 0, 256, relu=True)
    self.aspp_concat, self.concat_projection = cat, proj

    # ---- decoder (model.py:268-393).
    ll, lh, lw, lc = low_level
    dh = scale_dimension(H, 1.0 / 4)
    dw_ = scale_dimension(Wd, 1.0 / 4)
    assert (lh, lw) == (dh, dw_), ((lh, lw), (dh, dw_))
    # (padding the concat's rows 304 -> 320 and the stem's im2col rows 28 -> 32 as well:
    # measured neutral, profiles/r06/ab_pad_level.txt -- left dense)
    dcat = self._empty(B, dh, dw_, 304)
    ldd = self._ld(dcat)

    def run_up(stream, proj=proj, dcat=dcat, ldd=ldd):
      _lib.check(lib.epos_resize_bilinear_f32(
          _ptr(proj), 256, _ptr(dcat), ldd, B, eh, ew, dh, dw_, 256, stream),
                 'decoder/resize')
    self._add('decoder/resize', run_up)
    self._set_expr(dcat, self._expr_of(proj) if (eh, ew) == (dh, dw_) else
                   'resize(%s,%dx%d)' % (self._expr_of(proj), dh, dw_), 0, 256)
    m_dec = B * dh * dw_
    w_kn, sc, bi = self._conv_params('decoder/feature_projection0', HEAD_BN_EPS)
    self._pointwise('decoder/feature_projection0', ll, 0, lc, m_dec, lc, w_kn,
                    sc, bi, dcat, 256, ldd, relu=True)
    # dcat = [bilinear resize of proj | feature projection]: an interpolation never
    # exceeds its input's absmax, so the concat is bounded by the two producers' slots
    pb, fb = self._bound_of(proj), self._bound_of(dcat)
    if pb is not None and fb is not None:
      self._set_bound(dcat, fb[0], pb[0])
    else:
      self._bounds.pop(id(dcat), None)
    self.decoder_concat = dcat[..., :304]
    x, c = dcat, 304
    for j in range(2):
      scope = 'decoder/decoder_conv%d' % j
      d, _, _ = self._depthwise(scope + '_depthwise', x, self._ld(x), dh, dw_, c, 1, 1,
                                scope + '_depthwise', HEAD_BN_EPS, False, True)
      w_kn, sc, bi = self._conv_params(scope + '_pointwise', HEAD_BN_EPS)
      y = self._empty(B, dh, dw_, 256)
      self._pointwise(scope + '_pointwise', d, 0, self._ld(d), m_dec, c, w_kn, sc, bi, y,
                      0, 256, relu=True)
      x, c = y, 256
    self.decoder_out = x
    self.out_h, self.out_w = dh, dw_

    # ---- logits (model.py:396-458), sorted(name) order (model.py:503).
    self.logits = {}
    grp = []
    self._n_trunk_ops = len(self.ops)      # everything before the logits layers
    for name, ch in sorted(W.outputs_to_num_channels(
        self.num_objs, self.num_frags).items()):
      wt = self.ckpt['logits/%s/weights' % name].reshape(256, ch)
      bs = self.ckpt['logits/%s/biases' % name]
      buf = self._empty(B, dh, dw_, ch)
      self._pointwise('logits/' + name, x, 0, 256, m_dec, 256, wt,
                      np.ones(ch, np.float32), bs, buf, 0, ch, relu=False,
                      group=grp, track_out=False)
      self.logits[name] = buf
    # the dense heads (413 MB at C2) are written with streaming stores: nobody re-reads them
    # soon, and as ordinary stores they sweep the 256 MB Infinity Cache clean of the other
    # images' working sets (same box: 424.6 / 422.4 vs 422.0 / 420.4 images/s, prediction of a
    # serial step 3.31 vs 3.35 ms; EPOS_HEAD_NT_STORES=0 turns it off)
    if os.environ.get('EPOS_HEAD_NT_STORES', '1') == '1':
      for g in grp:
        g[1].c_stream = 1
    obj_only = [g for g in grp if g[0].endswith(W.PRED_OBJ_CONF)]
    # (Round 4 built the fragment softmax of model.py:678 into this launch's epilogue --
    # identical bits -- and measured it slower than the softmax's own memory-bound launch:
    # 410.6 / 415.6 vs 419.2 / 421.3 images/s, profiles/r04/ab_head_softmax.txt; removed in
    # round 5.)
    self._flush_group(grp)              # the three heads: one grouped launch
    # Sparse-head mode (pipeline option): only the object head runs densely; the
    # fragment heads are evaluated per (image, target object) -- see
    # run_sparse_heads().
    oname, oargs, oflops = obj_only[0][:3]

    def run_obj_head(stream, args=oargs):
      _lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(args), stream), oname)
    self._obj_head_op = (oname, run_obj_head)
    self._obj_head_flops = oflops
    self._sparse_packs = None
    self._decoder_x = x

    # ---- predict post-ops (model.py:677-683): softmax in place, argmax.
    self.post_ops = []
    obj = self.logits[W.PRED_OBJ_CONF]
    frag = self.logits[W.PRED_FRAG_CONF]
    self.obj_label = self._empty(B, dh, dw_, dtype=torch.int64)
    O, F = self.num_objs, self.num_frags
    if F > 64:
      raise ValueError('num_frags > 64 is not supported by the HIP softmax.')

    def run_softmax_obj(stream):
      _lib.check(lib.epos_softmax_groups_f32(_ptr(obj), m_dec, O + 1, stream),
                 'softmax_obj')

    def run_softmax_frag(stream):
      _lib.check(lib.epos_softmax_groups_f32(_ptr(frag), m_dec * O, F, stream),
                 'softmax_frag')

    def run_argmax(stream):
      _lib.check(lib.epos_argmax_i64(_ptr(obj), O + 1, _ptr(self.obj_label),
                                     m_dec, O + 1, stream), 'argmax')
    self.post_ops = [('softmax_obj', run_softmax_obj),
                     ('softmax_frag', run_softmax_frag),
                     ('argmax', run_argmax)]
    # model.py:117-147 (reshape), :677-683 (softmax over the last axis, argmax)
    eo = self._expr_of(self.logits[W.PRED_OBJ_CONF])
    ef = self._expr_of(self.logits[W.PRED_FRAG_CONF])
    el = self._expr_of(self.logits[W.PRED_FRAG_LOC])
    self.trace_outputs = {
        W.PRED_OBJ_CONF: {'expr': 'softmax(%s)' % eo, 'shape': [B, dh, dw_, O + 1]},
        W.PRED_OBJ_LABEL: {'expr': 'argmax(softmax(%s))' % eo, 'shape': [B, dh, dw_]},
        W.PRED_FRAG_CONF: {'expr': 'softmax(reshape(%s,%s))' % (ef, [O, F]),
                           'shape': [B, dh, dw_, O, F]},
        W.PRED_FRAG_LOC: {'expr': 'reshape(%s,%s)' % (el, [O, F, 3]),
                          'shape': [B, dh, dw_, O, F, 3]}}
    # The slot table is zeroed by the plan's first launch. Round 4: that is the im2col of the
    # first stem conv itself (EposIm2colArgs.amax_clear) when it directly follows -- one
    # launch less per image (EPOS_AMAX_CLEAR_FOLD=0 keeps the separate kernel).
    fi = getattr(self, '_first_im2col', None)
    if (not self.dry_run and fi is not None and len(self.ops) > 1 and
        self.ops[0][0] == 'amax_clear' and self.ops[1][0] == fi[0] and
        os.environ.get('EPOS_AMAX_CLEAR_FOLD', '1') == '1'):
      fi[1].amax_clear = _ptr(self._amax_table)
      fi[1].amax_words = self._n_slots * _lib.AMAX_WORDS
      del self.ops[0]
      self._n_trunk_ops -= 1

  def algorithmic_bytes(self, dense_heads=True):
    """HBM bytes of ONE pass of the plan under the fusion-group rule of SURVEY.md App. A
    (fp32): every separable conv is one group -- the depthwise INPUT is read once, the
    pointwise output written once, weights once, the intermediate never leaves the chip --
    a GEMM reads A and its residual once and writes its output once, the stem's im2col
    matrix does not exist (the input image is read once), and the element-wise ops that
    belong in a producer's epilogue (global mean, decoder resize, softmax, argmax) move
    nothing of their own: the heads are written once. ResNet's max-pool / subsample /
    stand-alone add read and write their tensors once. Reproduces the survey's figures
    (C2: 3.32 vs 3.30 GB, batch 8: 3.17 vs 3.15, C4: 4.40 vs 4.38; C5: 3.64 GB per image at batch 8). What the launches of
    this build actually move is roofline.traffic in bench.py."""
    total = 0
    for name, (a_bytes, rest, a_id) in self.op_io.items():
      if name.endswith('/im2col'):
        continue
      if a_id in self._dw_reads:            # A = a depthwise output: the group reads its input
        total += self._dw_reads[a_id] + rest
      elif name.endswith('conv1_1') and self.op_kind.get(name + '/im2col') == 'im2col':
        total += 4 * self.B * self.H * self.W * 3 + rest     # the image, not the im2col matrix
      else:
        total += a_bytes + rest
    B, h, w = self.B, self.out_h, self.out_w
    O, F = self.num_objs, self.num_frags
    total += getattr(self, '_glue_bytes', 0)
    if not dense_heads:
      total -= 4 * B * h * w * (O * F + 3 * O * F)
    return total

  # ----------------------------------------------------------- running ---
  def _stream(self):
    return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

  def sync_current(self):
    torch.cuda.current_stream(self.dev).synchronize()

  def run_plan(self, with_post=True, sparse=False, skip_kinds=()):
    """sparse=False: the whole dense plan. sparse=True: trunk + object head +
    object softmax/argmax only (the fragment heads follow per target object).
    skip_kinds: op kinds left out (capture_alt: the plan WITHOUT its GEMM / depthwise
    launches, for the step decomposition of bench.py). 'post' leaves out the in-place
    softmax / argmax post-ops: without the GEMMs the head buffers keep the probabilities
    of the last full run, and softmax applied to them again and again would flatten them
    (the stages downstream would then see no -- or, with one object, all -- pixels)."""
    s = self._stream()
    if not sparse:
      for name, fn in self.ops:
        if self.op_kind.get(name) not in skip_kinds:
          fn(s)
      if with_post and 'post' not in skip_kinds:
        for name, fn in self.post_ops:
          fn(s)
      return
    for name, fn in self.ops[:self._n_trunk_ops]:
      fn(s)
    self._obj_head_op[1](s)
    for name, fn in self.post_ops:
      if name != 'softmax_frag':
        fn(s)

  # ------------------------------------------------------- sparse heads ---
  def _build_sparse_packs(self):
    O, F = self.num_objs, self.num_frags
    wc = self.ckpt['logits/%s/weights' % W.PRED_FRAG_CONF].reshape(256, O * F)
    bc = self.ckpt['logits/%s/biases' % W.PRED_FRAG_CONF]
    wl = self.ckpt['logits/%s/weights' % W.PRED_FRAG_LOC].reshape(256, O * F * 3)
    bl = self.ckpt['logits/%s/biases' % W.PRED_FRAG_LOC]
    packs = []
    for o in range(O):
      one_c, one_l = np.ones(F, np.float32), np.ones(3 * F, np.float32)
      pc = self._pack_pointwise(wc[:, o * F:(o + 1) * F], one_c,
                                bc[o * F:(o + 1) * F])
      pc = pc + (self._pack_split(wc[:, o * F:(o + 1) * F], one_c),
                 self._pack_h2(wc[:, o * F:(o + 1) * F], one_c))
      pl = self._pack_pointwise(wl[:, o * 3 * F:(o + 1) * 3 * F], one_l,
                                bl[o * 3 * F:(o + 1) * 3 * F])
      pl = pl + (self._pack_split(wl[:, o * 3 * F:(o + 1) * 3 * F], one_l),
                 self._pack_h2(wl[:, o * 3 * F:(o + 1) * 3 * F], one_l))
      packs.append((pc, pl))
    self._sparse_packs = packs

  def run_sparse_heads(self, slots, slots_dev):
    """Fragment heads (model.py:449-456) + fragment softmax (model.py:678) for the
    given (image, obj_id) slots only: per slot one 256 -> F and one 256 -> 3F GEMM
    writing the object's channel slice of the dense head buffers, as grouped
    launches. Channels of other objects are left untouched (never read: the
    correspondence stage only visits the slots, corresp.py:42-43).
    slots_dev: int32 [S,2] device copy of ``slots`` for the softmax kernel."""
    if self._sparse_packs is None:
      self._build_sparse_packs()
    O, F = self.num_objs, self.num_frags
    P = self.out_h * self.out_w
    x = self._decoder_x
    conf, loc = self.logits[W.PRED_FRAG_CONF], self.logits[W.PRED_FRAG_LOC]
    s = self._stream()
    lib = self.lib
    flops = 0
    for kind in (0, 1):
      probs = []
      for im, obj_id in slots:
        wp, bp, _, ws, wh = self._sparse_packs[obj_id - 1][kind]
        xb = self._bound_of(x)
        if xb is None:
          wh = None
        n = F if kind == 0 else 3 * F
        buf = conf if kind == 0 else loc
        ldc = O * n
        probs.append(_lib.PointwiseArgs(
            A=_ptr(x, im * P * 256), lda=256, Wp=_ptr(wp), bias=_ptr(bp), R=None,
            ldr=0, C=_ptr(buf, im * P * ldc + (obj_id - 1) * n), ldc=ldc, M=P,
            N=n, K=256, relu=0, relu_in=0, sub=1, Ws=_ptr(ws),
            Wh=_ptr(wh) if wh is not None else None,
            a_amax=self._slot_ptr(xb[0]) if wh is not None else None,
            a_amax2=self._slot_ptr(xb[1]) if wh is not None else None,
            a_gain=xb[2] if wh is not None else 0.0,
            a_bias=xb[3] if wh is not None else 0.0))
        flops += 2 * P * n * 256
      for i in range(0, len(probs), 8):
        chunk = probs[i:i + 8]
        arr = (_lib.PointwiseArgs * len(chunk))(*chunk)
        _lib.check(lib.epos_pointwise_conv_grouped_f32(arr, len(chunk), s), 'sparse heads')
    if slots:
      _lib.check(lib.epos_softmax_slots_f32(_ptr(conf), _ptr(slots_dev),
                                            len(slots), P, O, F, s),
                 'softmax_slots')
    return flops

  def set_images(self, images):
    """images: [B,H,W,3] in [0,255] (host numpy or device tensor), float32 -- or uint8 as the
    decoder delivers them (datagen.py:435-436 casts to float32 right behind decode_image; here
    the bytes are uploaded and cast on the device, epos_u8_to_f32). Enqueued on the current
    stream; a pinned host tensor is uploaded without blocking the host."""
    t = torch.as_tensor(images)
    if t.dtype == torch.uint8:
      n = self.B * self.H * self.W * 3
      t = t.reshape(-1)
      if t.numel() != n:
        raise ValueError('images: expected %d values, got %d' % (n, t.numel()))
      if not t.is_cuda and t.is_pinned() and t.is_contiguous() and t.data_ptr() % 16 == 0 and \
          os.environ.get('EPOS_UPLOAD', 'zerocopy') == 'zerocopy':
        # pinned host memory is mapped into the device's address space: the cast kernel reads
        # the bytes over PCIe itself (0.9 MB per 640 x 480 frame) -- no copy engine, no
        # cross-engine dependency in front of the step's first kernel. The caller keeps the
        # buffer untouched until the step has been collected. EPOS_UPLOAD=copy: H2D copy first.
        src = t
      else:
        if self._images_u8 is None:
          self._images_u8 = torch.empty(n, dtype=torch.uint8, device=self.dev)
        self._images_u8.copy_(t, non_blocking=True)
        src = self._images_u8
      _lib.check(self.lib.epos_u8_to_f32(_ptr(src), _ptr(self.images), n, self._stream()),
                 'u8_to_f32')
      return
    if t.dtype != torch.float32:
      if t.is_cuda:
        raise TypeError('device images must be float32 or uint8, got %s' % t.dtype)
      t = t.float()
    self.images.copy_(t.reshape(self.B, self.H, self.W, 3), non_blocking=True)

  def capture_graph(self, sparse=False):
    """Captures the plan (dense, or the sparse-mode trunk) into one hipGraph."""
    torch.cuda.synchronize(self.dev)
    # Capture on the caller's stream when it is not the default one (the pipeline's
    # own stream): every extra HIP stream shifts the stream -> hardware-queue mapping
    # (4 queues), and two pipelines sharing a queue serialise against each other.
    side = torch.cuda.current_stream(self.dev)
    if side == torch.cuda.default_stream(self.dev):
      side = _capture_stream(self.dev)
    with torch.cuda.stream(side):
      self.run_plan(sparse=sparse)         # warm-up outside capture
    torch.cuda.synchronize(self.dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
      self.run_plan(sparse=sparse)
    if sparse:
      self._graph_sparse = g
    else:
      self._graph = g
    return g

  def capture_alt(self, skip_kinds):
    """A second hipGraph of the dense plan without the launches of the given kinds
    ('gemm', 'dw'); forward(use_graph=True) replays it while `alt_skip` is set. bench.py
    times the same pipelined steps with it: step time - that time = what the left-out
    kernels cost INSIDE the timed regime (overlap with the other plans included)."""
    if skip_kinds is None:
      self._graph_alt, self.alt_skip = None, None
      return
    torch.cuda.synchronize(self.dev)
    side = torch.cuda.current_stream(self.dev)
    if side == torch.cuda.default_stream(self.dev):
      side = _capture_stream(self.dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
      self.run_plan(skip_kinds=tuple(skip_kinds))
    self._graph_alt, self.alt_skip = g, tuple(skip_kinds)

  def forward(self, images=None, use_graph=False, sparse=False):
    """Runs the plan (logits + softmax/argmax post-ops) on the current stream and
    returns the prediction dict of ``model.predict`` (model.py:629-687) as views
    of the plan's HBM buffers (valid until the next forward)."""
    if images is not None:
      self.set_images(images)
    if use_graph:
      if sparse:
        if self._graph_sparse is None:
          self.capture_graph(sparse=True)
        self._graph_sparse.replay()
      elif self._graph_alt is not None:
        self._graph_alt.replay()
      else:
        if self._graph is None:
          self.capture_graph()
        self._graph.replay()
    else:
      self.run_plan(sparse=sparse)
    B, h, w = self.B, self.out_h, self.out_w
    O, F = self.num_objs, self.num_frags
    return {
        W.PRED_OBJ_CONF: self.logits[W.PRED_OBJ_CONF],
        W.PRED_OBJ_LABEL: self.obj_label,
        W.PRED_FRAG_CONF: self.logits[W.PRED_FRAG_CONF].view(B, h, w, O, F),
        W.PRED_FRAG_LOC: self.logits[W.PRED_FRAG_LOC].view(B, h, w, O, F, 3),
    }

  def outputs(self):
    """The prediction dict of the LAST dense run as views of the plan's HBM buffers (no
    launch): what forward() returned, for callers that read the heads after a pipeline
    step (infer.py --vis / --save_corresp)."""
    B, h, w = self.B, self.out_h, self.out_w
    O, F = self.num_objs, self.num_frags
    return {
        W.PRED_OBJ_CONF: self.logits[W.PRED_OBJ_CONF],
        W.PRED_OBJ_LABEL: self.obj_label,
        W.PRED_FRAG_CONF: self.logits[W.PRED_FRAG_CONF].view(B, h, w, O, F),
        W.PRED_FRAG_LOC: self.logits[W.PRED_FRAG_LOC].view(B, h, w, O, F, 3),
    }

  def time_ops(self, iters=3, warm=0):
    """Per-launch HIP-event timing of the plan (diagnostics). `warm` extra
    untimed launches per op let the core clock ramp (2.06 -> 2.4 GHz)."""
    out = []
    s = self._stream()
    for name, fn in self.ops:
      for _ in range(1 + warm):
        fn(s)
      torch.cuda.synchronize(self.dev)
      e0 = torch.cuda.Event(enable_timing=True)
      e1 = torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(iters):
        fn(s)
      e1.record()
      torch.cuda.synchronize(self.dev)
      out.append((name, e0.elapsed_time(e1) / iters, self.op_flops.get(name, 0)))
    return out
