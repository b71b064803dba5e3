"""GPU tests of the fp16-pair GEMM (pointwise_gemm_h2_f32, round 3: two fp16 pieces per
operand, three piece products per fp32 product -- the default fp32 GEMM), through the C ABI.
Same bars as the bf16 x 6 split kernel it replaces: tests/test_gpu_layers.py::
test_pointwise_gemm_split_ring / _accuracy / test_split_gemm_adversarial_operands."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -24


@pytest.fixture(scope='module')
def lib():
  """The TEST build of the library (libepos_hip_ref.so: the product library's own objects +
  the fp32-MFMA reference GEMM kernels of csrc/ref/): this module runs the product kernels
  against the fp32-MFMA kernel side by side, which the product library no longer carries."""
  from epos_amd import _lib
  assert torch.cuda.is_available(), 'GPU tests need a HIP device'
  return _lib.load_ref()


def _check(rc, what=''):
  from epos_amd import _lib
  return _lib.check(rc, what, lib=_lib.load_ref())


@pytest.fixture(autouse=True, params=['tile128', 'tile64'])
def tile(request, lib):
  """Every test of this module runs with both launch shapes of the fp16-pair kernel: 128 x 128
  tiles only (limit 0) and 128 x 64 tiles wherever the kernel offers them
  (epos_set_h2_narrow_tile_limit, include/epos_hip.h)."""
  prev = lib.epos_set_h2_narrow_tile_limit((1 << 30) if request.param == 'tile64' else 0)
  yield request.param
  lib.epos_set_h2_narrow_tile_limit(prev)


def _p(t, off=0):
  return ctypes.c_void_p(t.data_ptr() + off * t.element_size())


def _pack(lib, w_kn, which='plain'):
  k, n = w_kn.shape
  w = np.ascontiguousarray(w_kn, np.float32)
  wp = w.ctypes.data_as(ctypes.c_void_p)
  fn = {'plain': lib.epos_pack_pointwise_weights, 'split': lib.epos_pack_pointwise_weights_split,
        'h2': lib.epos_pack_pointwise_weights_h2}[which]
  total = fn(wp, k, n, None)
  if total <= 0:
    return None
  dst = np.empty(total, np.float32 if which == 'plain' else np.uint8)
  fn(wp, k, n, dst.ctypes.data_as(ctypes.c_void_p))
  return torch.from_numpy(dst).cuda()


def _slot():
  from epos_amd import _lib
  return torch.zeros(_lib.AMAX_WORDS, dtype=torch.int32, device='cuda')


def _slot_value(slot):
  return float(slot.cpu().numpy().view(np.float32).max())


def _gemm(lib, a, w, kind, bias=None, res=None, relu=0, a_amax=None, a_gain=0.0, a_bias=0.0,
          c_amax=None, lda=None):
  """C = A W through epos_pointwise_conv_f32; kind: 'fp32' | 'split' | 'h2'."""
  from epos_amd import _lib
  m, k = a.shape
  n = w.shape[1]
  lda = lda or k
  abuf = np.full((m, lda), np.nan, np.float32)
  abuf[:, :k] = a
  A = torch.from_numpy(abuf).cuda()
  C = torch.zeros(m, n, device='cuda')
  keep = [A, C, _pack(lib, w)]
  ws = _pack(lib, w, 'split') if kind in ('split', 'h2') else None
  wh = _pack(lib, w, 'h2') if kind == 'h2' else None
  assert kind != 'h2' or wh is not None, 'the packer refused this matrix'
  bd = rd = None
  if bias is not None:
    bp = np.zeros((n + 127) // 128 * 128, np.float32); bp[:n] = bias
    bd = torch.from_numpy(bp).cuda()
  if res is not None:
    rd = torch.from_numpy(np.ascontiguousarray(res, np.float32)).cuda()
  args = _lib.PointwiseArgs(A=_p(A), lda=lda, Wp=_p(keep[2]), bias=_p(bd) if bd is not None else None,
                            R=_p(rd) if rd is not None else None, ldr=n, C=_p(C), ldc=n,
                            M=m, N=n, K=k, relu=relu, relu_in=0, sub=1,
                            Ws=_p(ws) if ws is not None else None,
                            Wh=_p(wh) if wh is not None else None,
                            a_amax=_p(a_amax) if a_amax is not None else None,
                            a_gain=a_gain, a_bias=a_bias,
                            c_amax=_p(c_amax) if c_amax is not None else None)
  _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
  torch.cuda.synchronize()
  return C.cpu().numpy()


@pytest.mark.parametrize('k', [16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 40, 92, 728])
@pytest.mark.parametrize('m,n,aligned,res', [(128, 128, 1, 0), (130, 200, 1, 1),
                                             (257, 132, 0, 1), (1000, 96, 1, 0),
                                             (16700, 500, 1, 1), (16450, 490, 0, 0)])
def test_pointwise_gemm_h2_ring(lib, k, m, n, aligned, res):
  """Every prologue / steady / tail path of the five-stage ring (1..46 K steps, partial last
  step), ragged M and N tiles (3 and 4 live column blocks), float4 and scalar epilogue,
  residual + ReLU, output into a wider buffer at an offset; NaNs behind every row of A
  poison any read past K (also in the library's own absmax pass, taken here because no
  slot is passed)."""
  from epos_amd import _lib
  rng = np.random.RandomState(k * 7 + m + n)
  a = rng.standard_normal((m, k)).astype(np.float32)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  bias = rng.standard_normal(n).astype(np.float32)
  r = rng.standard_normal((m, n)).astype(np.float32)
  npad = (n + 127) // 128 * 128
  bpad = np.zeros(npad, np.float32); bpad[:n] = bias
  lda = k + 36
  abuf = np.full((m, lda), np.nan, np.float32); abuf[:, :k] = a
  A, Wp, Wh, Bd, R = (torch.from_numpy(abuf).cuda(), _pack(lib, w), _pack(lib, w, 'h2'),
                      torch.from_numpy(bpad).cuda(), torch.from_numpy(r).cuda())
  ldc = n + (4 if aligned else 5)
  off = 4 if aligned else 3
  C = torch.full((m, ldc), -7.0, device='cuda')
  args = _lib.PointwiseArgs(A=_p(A), lda=lda, Wp=_p(Wp), bias=_p(Bd),
                            R=_p(R) if res else None, ldr=n, C=_p(C, off), ldc=ldc,
                            M=m, N=n, K=k, relu=res, relu_in=0, sub=1, Wh=_p(Wh))
  _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
  torch.cuda.synchronize()
  out = C.cpu().numpy()
  ref = a.astype(np.float64) @ w.astype(np.float64) + bias
  if res:
    ref = np.maximum(ref + r, 0)
  np.testing.assert_allclose(out[:, off:off + n], ref, rtol=2e-5, atol=2e-5)
  assert (out[:, :off] == -7.0).all() and (out[:, off + n:] == -7.0).all()


@pytest.mark.parametrize('m,k,n', [(4800, 728, 728), (2048, 2048, 256), (4096, 256, 1344)])
def test_pointwise_gemm_h2_accuracy(lib, m, k, n):
  """The claim the kernel rests on (same assertions as test_pointwise_gemm_split_accuracy):
  its error against an fp64 product is NOT larger than the fp32-MFMA kernel's on the same
  inputs (ReLU-like activations, as in the network), relative to sum_k |a||w| per output."""
  rng = np.random.RandomState(k + n)
  a = np.maximum(rng.standard_normal((m, k)), 0).astype(np.float32)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  ref = a.astype(np.float64) @ w.astype(np.float64)
  mag = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64)
  errs = {}
  for kind in ('fp32', 'split', 'h2'):
    c = _gemm(lib, a, w, kind).astype(np.float64)
    e = np.abs(c - ref) / mag
    errs[kind] = (float(np.sqrt((e * e).mean())), float(e.max()))
  print('rms / max error relative to sum|a||w|:', errs)
  assert errs['h2'][0] <= errs['fp32'][0] * 1.05
  assert errs['h2'][1] <= errs['fp32'][1] * 1.5
  assert errs['h2'][1] < 4e-7


def _heavy_tailed(rng, k, n, tails):
  """Weight matrices with the tails a trained checkpoint has and random init has not
  (VERDICT r05 weak #3): log-normal sigma = 3 magnitudes; a Gaussian matrix with 1 % of its
  entries scaled by 1e-10; Gaussian with ONE weight at 1e-12 of its column maximum."""
  g = rng.standard_normal((k, n)) / np.sqrt(k)
  if tails == 'lognormal3':
    w = np.exp(3.0 * rng.standard_normal((k, n))) * rng.choice([-1.0, 1.0], (k, n))
    w /= np.sqrt((w * w).sum(0, keepdims=True))         # unit columns, like g
  elif tails == 'sparse1e-10':
    w = g.copy()
    w[rng.uniform(size=w.shape) < 0.01] *= 1e-10
  elif tails == 'one1e-12':
    w = g.copy()
    w[5, :] = 1e-12 * np.abs(g).max(0)
  else:
    raise ValueError(tails)
  return w.astype(np.float32)


@pytest.mark.parametrize('tails', ['lognormal3', 'sparse1e-10', 'one1e-12'])
@pytest.mark.parametrize('m,k,n', [(4800, 728, 728), (4096, 256, 1344)])
def test_pointwise_gemm_h2_heavy_tailed_weights(lib, m, k, n, tails):
  """Round 6: weight matrices the packer used to refuse (one weight more than ~2^27 below its
  column's maximum) now run on the fp16-pair kernel -- a middle-flow and a heads-shaped GEMM
  against the fp64 product, held to the SAME bars as test_pointwise_gemm_h2_accuracy:
  rms and maximum error relative to sum_k |a||w| not above the fp32-MFMA kernel's."""
  rng = np.random.RandomState(k + n + len(tails))
  a = np.maximum(rng.standard_normal((m, k)), 0).astype(np.float32)
  w = _heavy_tailed(rng, k, n, tails)
  assert _pack(lib, w, 'h2') is not None
  ref = a.astype(np.float64) @ w.astype(np.float64)
  mag = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64)
  errs = {}
  for kind in ('fp32', 'h2'):
    c = _gemm(lib, a, w, kind).astype(np.float64)
    e = np.abs(c - ref) / mag
    errs[kind] = (float(np.sqrt((e * e).mean())), float(e.max()))
  print(tails, 'rms / max error relative to sum|a||w|:', errs)
  assert errs['h2'][0] <= errs['fp32'][0] * 1.05
  assert errs['h2'][1] <= errs['fp32'][1] * 1.5
  # the absolute bar of the Gaussian case holds where the column's terms are of one size; a
  # log-normal column is a few dominant terms plus hundreds of small ones that fp32
  # ACCUMULATION absorbs (the fp32-MFMA kernel itself reaches 1.9e-6 there, the fp16-pair
  # kernel 1.2e-6: measured, profiles/r06/h2_heavy_tails.txt) -- there the bar is the
  # fp32 kernel's own error
  assert errs['h2'][1] < (4e-7 if tails != 'lognormal3' else max(4e-7, errs['fp32'][1]))


def test_h2_subnormal_weight_pieces_reach_the_matrix_pipe(lib):
  """The graceful side of the packer's contract, isolated: every column has ONE dominant
  weight (it fixes the column's scale) whose activation is exactly zero, so the result is made
  of the tiny weights alone -- 2^-30 .. 2^-44 of the column maximum, i.e. fp16 SUBNORMAL hi /
  mid pieces. If the matrix pipe flushed them the result would be 0; the contract says
  |c - ref| <= sum_k |a_k| x max(2^-22 |w_k|, 2^-50 x column maximum) (+ the dropped
  mid x mid products and the fp32 accumulation, 2^-21 of sum|a||w|)."""
  rng = np.random.RandomState(9)
  m, k, n = 256, 128, 128
  a = rng.uniform(0.5, 1.0, (m, k)).astype(np.float32)
  a[:, 0] = 0
  w = (rng.uniform(1, 2, (k, n)) * 2.0 ** rng.randint(-44, -29, (k, n)) *
       rng.choice([-1, 1], (k, n))).astype(np.float32)
  w[0, :] = rng.uniform(1, 2, n).astype(np.float32)
  slot = _slot()
  A = torch.from_numpy(a).cuda()
  from epos_amd import _lib
  _check(lib.epos_absmax_f32(_p(A), k, m, k, _p(slot), None))
  c = _gemm(lib, a, w, 'h2', a_amax=slot).astype(np.float64)
  ref = a.astype(np.float64) @ w.astype(np.float64)
  assert np.abs(ref).min() > 0 and (c != 0).all()
  cmax = np.abs(w).astype(np.float64).max(0)
  werr = np.maximum(np.abs(w).astype(np.float64) * 2.0 ** -22, cmax[None, :] * 2.0 ** -50)
  bound = np.abs(a).astype(np.float64) @ werr + \
      2.0 ** -21 * (np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64))
  assert (np.abs(c - ref) <= bound).all(), float((np.abs(c - ref) / bound).max())
  # and the weights inside the window (2^-30 .. 2^-28 is still outside; use 2^-27 .. 2^-20):
  w2 = (rng.uniform(1, 2, (k, n)) * 2.0 ** rng.randint(-27, -19, (k, n)) *
        rng.choice([-1, 1], (k, n))).astype(np.float32)
  w2[0, :] = w[0, :]
  c2 = _gemm(lib, a, w2, 'h2', a_amax=slot).astype(np.float64)
  ref2 = a.astype(np.float64) @ w2.astype(np.float64)
  mag2 = np.abs(a).astype(np.float64) @ np.abs(w2).astype(np.float64)
  assert (np.abs(c2 - ref2) <= 4 * EPS * mag2).all()


def test_h2_result_does_not_depend_on_the_bound(lib):
  """The power of two only moves exponents: as long as every element stays inside the
  window of full precision (~2^27 below the bound) a looser bound (slot x gain + bias)
  gives the same bits; a bound 1000x too large costs 10 octaves of that window and only the
  smallest elements (< 2^-17 of the maximum here) lose relative precision -- absolute error
  <= 2^-50 x bound each."""
  rng = np.random.RandomState(5)
  m, k, n = 700, 328, 260
  a = np.maximum(rng.standard_normal((m, k)), 0).astype(np.float32)
  a[(a > 0) & (a < 1e-3)] = 1e-3       # ReLU outputs, none closer to 0 than 2^-12 of the max
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  base = _gemm(lib, a, w, 'h2')
  from epos_amd import _lib
  for gain, bias in ((0.0, 0.0), (8.0, 0.0), (1.0, 3.7), (1000.0, 50.0)):
    slot = _slot()
    A = torch.from_numpy(a).cuda()
    _check(lib.epos_absmax_f32(_p(A), k, m, k, _p(slot), None))
    torch.cuda.synchronize()
    assert _slot_value(slot) == np.abs(a).max()
    c = _gemm(lib, a, w, 'h2', a_amax=slot, a_gain=gain, a_bias=bias)
    assert np.array_equal(c, base), (gain, bias)
  # elements below the shrunken window: still within the absolute floor
  a2 = a.copy()
  a2[:, ::7] = np.float32(2.0 ** -22) * np.maximum(rng.standard_normal((m, (k + 6) // 7)), 0)
  ref = a2.astype(np.float64) @ w.astype(np.float64)
  slot = _slot()
  A = torch.from_numpy(a2).cuda()
  _check(lib.epos_absmax_f32(_p(A), k, m, k, _p(slot), None))
  c = _gemm(lib, a2, w, 'h2', a_amax=slot, a_gain=1000.0, a_bias=50.0)
  bound = 1000.0 * np.abs(a2).max() + 50.0
  floor = 2.0 ** -50 * 2 * bound * np.abs(w).astype(np.float64).sum(0)
  mag = np.abs(a2).astype(np.float64) @ np.abs(w).astype(np.float64)
  assert (np.abs(c - ref) <= 4 * EPS * mag + floor[None, :]).all()


def test_h2_publishes_the_output_absmax(lib):
  """c_amax receives max|C| over what the launch wrote (bias, residual and ReLU applied),
  from the h2, the bf16 x 6 and the fp32-MFMA kernels alike; it accumulates by max."""
  rng = np.random.RandomState(6)
  m, k, n = 1000, 96, 392
  a = rng.standard_normal((m, k)).astype(np.float32)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  bias = rng.standard_normal(n).astype(np.float32)
  r = rng.standard_normal((m, n)).astype(np.float32)
  for kind in ('h2', 'split', 'fp32'):
    for relu in (0, 1):
      slot = _slot()
      c = _gemm(lib, a, w, kind, bias=bias, res=r, relu=relu, c_amax=slot)
      assert _slot_value(slot) == np.abs(c).max(), (kind, relu)
      c2 = _gemm(lib, 0.5 * a, w, kind, c_amax=slot)          # smaller: the slot stays
      assert _slot_value(slot) == max(np.abs(c).max(), np.abs(c2).max())


def test_absmax_kernel(lib):
  from epos_amd import _lib
  rng = np.random.RandomState(7)
  for rows, cols, ldx in ((1, 256, 256), (4800, 728, 728), (1000, 30, 36), (77, 5, 5)):
    x = rng.standard_normal((rows, ldx)).astype(np.float32)
    x[:, cols:] = 1e9                                        # outside the window
    X = torch.from_numpy(x).cuda()
    slot = _slot()
    _check(lib.epos_absmax_f32(_p(X), ldx, rows, cols, _p(slot), None))
    torch.cuda.synchronize()
    assert _slot_value(slot) == np.abs(x[:, :cols]).max()
    _check(lib.epos_amax_clear(_p(slot), 1, None))
    torch.cuda.synchronize()
    assert _slot_value(slot) == 0.0


@pytest.mark.parametrize('m,k,n,res,relu', [(4800, 728, 728, 0, 0), (4800, 728, 728, 1, 1),
                                            (1200, 1024, 1536, 0, 1), (333, 40, 24, 0, 0),
                                            (19200, 256, 48, 0, 1), (130, 92, 200, 1, 0)])
def test_h2_bits_do_not_depend_on_the_tile(lib, m, k, n, res, relu):
  """128 x 64 and 128 x 128 tiles accumulate an element's K sum in the same order: equal bits,
  equal published absmax -- with and without residual, ragged M / N / K, the first and the
  second half of a packed 128-column weight image."""
  from epos_amd import _lib
  rng = np.random.RandomState(m + k + n)
  a = rng.standard_normal((m, k)).astype(np.float32)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  bias = rng.standard_normal(n).astype(np.float32)
  r = rng.standard_normal((m, n)).astype(np.float32) if res else None
  outs = []
  prev = lib.epos_set_h2_narrow_tile_limit(0)
  try:
    for limit in (0, 1 << 30):
      lib.epos_set_h2_narrow_tile_limit(limit)
      slot_a, slot_c = _slot(), _slot()
      A = torch.from_numpy(a).cuda()
      _check(lib.epos_absmax_f32(_p(A), k, m, k, _p(slot_a), None))
      c = _gemm(lib, a, w, 'h2', bias=bias, res=r, relu=relu, a_amax=slot_a, c_amax=slot_c)
      outs.append((c, _slot_value(slot_c)))
  finally:
    lib.epos_set_h2_narrow_tile_limit(prev)
  for o in outs[1:]:
    assert np.array_equal(outs[0][0].view(np.uint32), o[0].view(np.uint32))
    assert o[1] == outs[0][1]
  assert outs[0][1] == np.abs(outs[0][0]).max()
  ref = a.astype(np.float64) @ w.astype(np.float64) + bias
  if res:
    ref = ref + r
  if relu:
    ref = np.maximum(ref, 0)
  np.testing.assert_allclose(outs[1][0], ref, rtol=2e-5, atol=2e-5)


def test_h2_grouped_and_strided(lib):
  """Three problems in one grid (one of them 22 columns wide: scalar epilogue inside a
  group) and a stride-2 row gather, through the h2 kernel."""
  from epos_amd import _lib
  rng = np.random.RandomState(12)
  b, hi, wi, cin = 2, 13, 18, 72
  ho, wo = (hi + 1) // 2, (wi + 1) // 2
  x = rng.standard_normal((b, hi, wi, cin)).astype(np.float32)
  X = torch.from_numpy(x).cuda()
  outs, refs, arr = [], [], (_lib.PointwiseArgs * 3)()
  keep = []
  for i, (n, sub) in enumerate([(136, 2), (22, 2), (260, 2)]):
    w = (rng.standard_normal((cin, n)) / np.sqrt(cin)).astype(np.float32)
    Wp, Wh = _pack(lib, w), _pack(lib, w, 'h2')
    C = torch.zeros(b * ho * wo, n, device='cuda')
    keep += [Wp, Wh, C]
    arr[i] = _lib.PointwiseArgs(A=_p(X), lda=cin, Wp=_p(Wp), bias=None, R=None, ldr=0,
                                C=_p(C), ldc=n, M=b * ho * wo, N=n, K=cin, relu=0,
                                relu_in=0, sub=sub, Ho=ho, Wo=wo, Hi=hi, Wi=wi,
                                Wh=_p(Wh))
    outs.append(C)
    refs.append(x[:, ::2, ::2, :].reshape(-1, cin).astype(np.float64) @ w)
  _check(lib.epos_pointwise_conv_grouped_f32(arr, 3, None))
  torch.cuda.synchronize()
  for C, ref in zip(outs, refs):
    np.testing.assert_allclose(C.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


def test_h2_group_with_residuals_equals_the_single_launches(lib):
  """A group whose problems carry residuals (round 5: the library has no grouped residual
  instantiation any more -- such a group goes out problem by problem) gives the bits of the
  same problems launched one by one; a group that mixes pre-split and fp32 A is refused."""
  from epos_amd import _lib
  rng = np.random.RandomState(5)
  m, k = 1000, 96
  a = rng.standard_normal((m, k)).astype(np.float32)
  A = torch.from_numpy(a).cuda()
  slot = _slot()
  _check(lib.epos_absmax_f32(_p(A), k, m, k, _p(slot), None))
  arr = (_lib.PointwiseArgs * 2)()
  keep, singles, grouped = [], [], []
  for i, n in enumerate((200, 64)):
    w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    r = torch.from_numpy(rng.standard_normal((m, n)).astype(np.float32)).cuda()
    Wp, Wh = _pack(lib, w), _pack(lib, w, 'h2')
    C1, C2 = torch.zeros(m, n, device='cuda'), torch.zeros(m, n, device='cuda')
    keep += [Wp, Wh, r]
    kw = dict(A=_p(A), lda=k, Wp=_p(Wp), bias=None, R=_p(r), ldr=n, ldc=n, M=m, N=n, K=k,
              relu=1, relu_in=0, sub=1, Wh=_p(Wh), a_amax=_p(slot))
    one = _lib.PointwiseArgs(C=_p(C1), **kw)
    _check(lib.epos_pointwise_conv_f32(ctypes.byref(one), None))
    arr[i] = _lib.PointwiseArgs(C=_p(C2), **kw)
    singles.append(C1); grouped.append(C2)
  _check(lib.epos_pointwise_conv_grouped_f32(arr, 2, None))
  torch.cuda.synchronize()
  for c1, c2 in zip(singles, grouped):
    assert torch.equal(c1, c2) and float(c1.abs().max()) > 0
  arr[1].a_presplit = 1                       # mixed kinds in one group
  assert lib.epos_pointwise_conv_grouped_f32(arr, 2, None) != 0


@pytest.mark.parametrize('b,h,w,cin,cout,stride,rate', [
    (2, 12, 16, 32, 64, 1, 1), (1, 9, 21, 64, 40, 1, 1), (2, 7, 5, 32, 136, 1, 1),
    (2, 15, 21, 64, 72, 2, 1), (1, 20, 28, 64, 130, 1, 2), (2, 13, 17, 32, 48, 1, 4),
    (4, 60, 80, 32, 64, 1, 1)])
def test_conv3x3_implicit_gemm_h2(lib, b, h, w, cin, cout, stride, rate):
  """Dense 3x3 conv as an implicit GEMM through the h2 kernel (taps gathered by the
  LDS-DMA, zero block outside the image) vs the oracle's conv2d_same."""
  from epos_amd import _lib
  from oracle import net_ref
  rng = np.random.RandomState(b * h + cin)
  x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
  wgt = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
  bias = rng.standard_normal(cout).astype(np.float32)
  xt = torch.from_numpy(x).permute(0, 3, 1, 2)
  if stride == 1:
    ref = net_ref.conv2d_raw(xt, wgt, 1, rate, 'SAME')
  else:
    ref = net_ref.conv2d_raw(net_ref.fixed_padding(xt, 3, rate), wgt, stride, rate, 'VALID')
  ref = np.maximum(ref.permute(0, 2, 3, 1).numpy() + bias, 0)
  ho, wo = ref.shape[1], ref.shape[2]
  X = torch.from_numpy(x).cuda()
  wkn = wgt.reshape(9 * cin, cout)
  Wp, Wh = _pack(lib, wkn), _pack(lib, wkn, 'h2')
  npad = (cout + 127) // 128 * 128
  bpad = np.zeros(npad, np.float32); bpad[:cout] = bias
  Bd = torch.from_numpy(bpad).cuda()
  Y = torch.full((b, ho, wo, cout), -3.0, device='cuda')
  xs, ys = _slot(), _slot()
  _check(lib.epos_absmax_f32(_p(X), cin, b * h * w, cin, _p(xs), None))
  args = _lib.Conv3x3Args(X=_p(X), ldx=cin, Wp=_p(Wp), bias=_p(Bd), Y=_p(Y), ldy=cout,
                          B=b, H=h, W=w, Cin=cin, Cout=cout, stride=stride, rate=rate,
                          relu=1, Wh=_p(Wh), x_amax=_p(xs),
                          y_amax=_p(ys) if cout % 4 == 0 else None)
  _check(lib.epos_conv3x3_f32(ctypes.byref(args), None))
  torch.cuda.synchronize()
  got = Y.cpu().numpy()
  np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
  if cout % 4 == 0:
    assert _slot_value(ys) == np.abs(got).max()


def test_h2_adversarial_operands(lib):
  """The fp32-equivalence claim outside the comfortable range, with the bounds of
  test_split_gemm_adversarial_operands (relative to sum_k |a_k||w_k|, eps = 2^-24) wherever
  fp16 pairs can hold the operands, and PROVABLE routing to the bf16 x 6 kernel where they
  cannot (the packer refuses the weights)."""
  rng = np.random.RandomState(0)
  m, k, n = 256, 512, 128
  rms = {}

  def rel_err(a, w):
    ref = a.astype(np.float64) @ w.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64)
    out = {}
    for kind in ('h2', 'fp32'):
      c = _gemm(lib, a, w, kind).astype(np.float64)
      with np.errstate(invalid='ignore', divide='ignore'):
        r = np.abs(c - ref) / np.maximum(scale, 1e-300)
      out[kind] = np.nanmax(r)
      rms[kind] = float(np.sqrt(np.nanmean(r * r)))
    return out

  # 1. alternating-sign cancellation: the exact result is tiny against the terms
  v = rng.uniform(1, 2, (m, k)).astype(np.float32)
  a = v * np.where(np.arange(k) % 2 == 0, 1.0, -1.0).astype(np.float32)
  a[:, 1::2] = -a[:, 0::2] * (1 + rng.uniform(-1e-6, 1e-6, (m, k // 2)).astype(np.float32))
  w = np.ones((k, n), np.float32) * rng.uniform(0.5, 1.5, (1, n)).astype(np.float32)
  e = rel_err(a, w)
  assert e['h2'] <= 2 * EPS and e['h2'] <= e['fp32'] * 1.01 + EPS / 8, e
  # 2. magnitudes spread over 2^-60 .. 2^60 in BOTH operands. Until round 5 the packer refused
  #    such a weight column; since round 6 it is accepted: weights more than 2^28 below their
  #    column's maximum are held to an ABSOLUTE 2^-50 x column maximum, which is far below
  #    sum|a||w| (dominated by the large pairs) -- the bound of the fp32 class holds
  w2 = (rng.uniform(1, 2, (k, n)) * 2.0 ** rng.randint(-60, 61, (k, n)) *
        rng.choice([-1, 1], (k, n))).astype(np.float32)
  assert _pack(lib, w2, 'h2') is not None
  a2 = (rng.uniform(1, 2, (m, k)) * 2.0 ** rng.randint(-60, 61, (m, k)) *
        rng.choice([-1, 1], (m, k))).astype(np.float32)
  e = rel_err(a2, w2)
  assert e['h2'] <= 16 * EPS and rms['h2'] <= 1.25 * rms['fp32'] + EPS / 8, (e, rms)
  # 2b. the same spread in A ONLY (per-tensor scale: elements more than ~2^27 below the
  #     tensor's maximum lose relative precision, their ABSOLUTE error stays below
  #     2^-50 x max|A| x |w|): with weights of one magnitude sum|a||w| is dominated by the
  #     large elements and the bound of the fp32 class holds
  a = (rng.uniform(1, 2, (m, k)) * 2.0 ** rng.randint(-60, 61, (m, k)) *
       rng.choice([-1, 1], (m, k))).astype(np.float32)
  w = (rng.uniform(1, 2, (k, n)) * rng.choice([-1, 1], (k, n))).astype(np.float32)
  e = rel_err(a, w)
  assert e['h2'] <= 16 * EPS and rms['h2'] <= 1.25 * rms['fp32'] + EPS / 8, (e, rms)
  # 2c. weights spread over the window the packer accepts (2^20 here) and activations
  #     spread over 2^20: every element is inside both windows
  a = (rng.uniform(1, 2, (m, k)) * 2.0 ** rng.randint(-10, 11, (m, k)) *
       rng.choice([-1, 1], (m, k))).astype(np.float32)
  w = (rng.uniform(1, 2, (k, n)) * 2.0 ** rng.randint(-10, 11, (k, n)) *
       rng.choice([-1, 1], (k, n))).astype(np.float32)
  e = rel_err(a, w)
  assert e['h2'] <= 16 * EPS and rms['h2'] <= 1.25 * rms['fp32'] + EPS / 8, (e, rms)
  # 3. tiny magnitudes (|x| ~ 2^-118 .. 2^-108): the scale 2^+122 is a normal fp32 number,
  #    the inverse scales are applied one after the other in the epilogue
  a = (rng.uniform(1, 2, (m, k)) * 2.0 ** rng.randint(-118, -107, (m, k))).astype(np.float32)
  w = rng.uniform(1, 2, (k, n)).astype(np.float32)
  e = rel_err(a, w)
  assert e['h2'] <= 32 * EPS and rms['h2'] <= 1.25 * rms['fp32'] + EPS / 8, (e, rms)
  #    and huge ones (2^100 .. 2^110)
  a = (rng.uniform(1, 2, (m, k)) * 2.0 ** rng.randint(100, 111, (m, k))).astype(np.float32)
  w = rng.uniform(2.0 ** -8, 2.0 ** -7, (k, n)).astype(np.float32)
  e = rel_err(a, w)
  assert e['h2'] <= 32 * EPS and rms['h2'] <= 1.25 * rms['fp32'] + EPS / 8, (e, rms)
  # 4. non-finite operands stay non-finite, finite rows are untouched (the bound is Inf /
  #    ignores NaN: scale 1 resp. the finite maximum)
  a = rng.standard_normal((m, k)).astype(np.float32)
  w = rng.standard_normal((k, n)).astype(np.float32)
  a[3, 17] = np.inf
  a[5, 100] = np.nan
  c = _gemm(lib, a, w, 'h2').astype(np.float64)
  assert not np.isfinite(c[3]).any() and np.isnan(c[5]).all()
  ok = np.ones(m, bool); ok[[3, 5]] = False
  ref = a[ok].astype(np.float64) @ w.astype(np.float64)
  assert np.abs(c[ok] - ref).max() <= 8 * EPS * (np.abs(a[ok]).astype(np.float64) @ np.abs(w)).max()
  # 5. an all-zero A and a zero column of W
  a = np.zeros((m, k), np.float32)
  w = rng.standard_normal((k, n)).astype(np.float32); w[:, 5] = 0
  assert (_gemm(lib, a, w, 'h2') == 0).all()
  a = rng.standard_normal((m, k)).astype(np.float32)
  c = _gemm(lib, a, w, 'h2')
  assert (c[:, 5] == 0).all() and np.abs(c).max() > 1


def test_h2_fp16_denormal_pieces_are_not_flushed(lib):
  """Elements 2^-30 of the tensor's maximum: their scaled value 2^-16 is an fp16 DENORMAL
  (8 significant bits in hi, the rest in a barely normal mid): the representation is good
  to ~2^-20 relative. If the matrix pipe flushed fp16 denormals these elements would vanish
  altogether (error of order 1)."""
  rng = np.random.RandomState(9)
  m, k, n = 256, 256, 128
  a = (rng.uniform(1, 2, (m, k)) * 2.0 ** -30).astype(np.float32)
  a[:, 0] = 1.0                         # the maximum that fixes the scale
  w = rng.uniform(1, 2, (k, n)).astype(np.float32)
  w[0] = 0                              # ... and contributes nothing
  ref = a.astype(np.float64) @ w.astype(np.float64)
  c = _gemm(lib, a, w, 'h2').astype(np.float64)
  assert np.abs(c - ref).max() <= 2.0 ** -18 * np.abs(ref).max()


@pytest.mark.parametrize('hi,wi,c,stride,rate,relu_in,relu_out', [
    (17, 23, 32, 1, 1, 0, 0), (60, 80, 728, 1, 2, 1, 0), (16, 24, 64, 2, 1, 0, 0),
    (30, 40, 256, 1, 12, 0, 1), (15, 21, 8, 2, 1, 1, 0), (24, 32, 304, 1, 1, 0, 1)])
def test_depthwise_fp16_pair_output_and_presplit_gemm(lib, hi, wi, c, stride, rate, relu_in,
                                                      relu_out):
  """EposDepthwiseArgs.y_h2: the depthwise kernel writes its output as fp16 pairs under the
  scale of its bound (gain * max|X| + max|bias|). (1) Decoded, the pairs reproduce the fp32
  output of the same kernel to 2^-22 relative (window of full precision). (2) The fp16-pair
  GEMM on that buffer (a_presplit) gives, BIT FOR BIT, what it gives on the fp32 buffer when
  it splits A itself with the same slots -- the split is the same arithmetic, done once."""
  from epos_amd import _lib
  rng = np.random.RandomState(hi * 7 + c)
  b = 2
  ho = hi if stride == 1 else (hi - 1) // 2 + 1
  wo = wi if stride == 1 else (wi - 1) // 2 + 1
  x = rng.standard_normal((b, hi, wi, c)).astype(np.float32)
  w9c = (rng.standard_normal((9, c)) * 0.4).astype(np.float32)
  bias = rng.standard_normal(c).astype(np.float32)
  X = torch.from_numpy(x).cuda()
  W9, Bd = torch.from_numpy(w9c).cuda(), torch.from_numpy(bias).cuda()
  xs = _slot()
  _check(lib.epos_absmax_f32(_p(X), c, b * hi * wi, c, _p(xs), None))
  gain = float(np.abs(w9c.astype(np.float64)).sum(0).max())
  bias0 = float(np.abs(bias).max())
  outs = {}
  for h2 in (0, 1):
    Y = torch.full((b, ho, wo, c), 7.0, device='cuda')
    args = _lib.DepthwiseArgs(X=_p(X), ldx=c, w9c=_p(W9), bias=_p(Bd), Y=_p(Y), ldy=c, B=b,
                              Hi=hi, Wi=wi, Ho=ho, Wo=wo, C=c, stride=stride, rate=rate,
                              relu_in=relu_in, relu_out=relu_out, y_h2=h2, x_amax=_p(xs),
                              gain=gain, bias0=bias0)
    _check(lib.epos_depthwise3x3_f32(ctypes.byref(args), None))
    torch.cuda.synchronize()
    outs[h2] = Y
  y32 = outs[0].cpu().numpy()
  raw = outs[1].cpu().numpy().view(np.float16).reshape(b, ho, wo, c // 4, 2, 4)
  hi16, mid16 = raw[..., 0, :].astype(np.float64), raw[..., 1, :].astype(np.float64)
  bound = gain * np.abs(x).max() + bias0
  assert np.abs(y32).max() <= bound                      # the bound holds
  s = 2.0 ** (14 - np.floor(np.log2(bound)))
  dec = ((hi16 + mid16 / 2048.0) / s).reshape(b, ho, wo, c)
  assert np.abs(hi16).max() < 2.0 ** 15
  big = np.abs(y32) >= 2.0 ** -26 * bound
  assert (np.abs(dec - y32)[big] <= 2.0 ** -22 * np.abs(y32)[big]).all()
  assert (np.abs(dec - y32)[~big] <= 2.0 ** -49 * bound).all()
  # (2) the GEMM on either form
  n = 136
  wk = (rng.standard_normal((c, n)) / np.sqrt(c)).astype(np.float32)
  Wp, Wh = _pack(lib, wk), _pack(lib, wk, 'h2')
  res = {}
  for h2 in (0, 1):
    C = torch.zeros(b * ho * wo, n, device='cuda')
    a = _lib.PointwiseArgs(A=_p(outs[h2]), lda=c, Wp=_p(Wp), bias=None, R=None, ldr=0, C=_p(C),
                           ldc=n, M=b * ho * wo, N=n, K=c, relu=0, relu_in=0, sub=1,
                           Wh=_p(Wh), a_amax=_p(xs), a_gain=gain, a_bias=bias0, a_presplit=h2)
    _check(lib.epos_pointwise_conv_f32(ctypes.byref(a), None))
    torch.cuda.synchronize()
    res[h2] = C.cpu().numpy()
  assert np.array_equal(res[0], res[1])
  ref = y32.reshape(-1, c).astype(np.float64) @ wk.astype(np.float64)
  np.testing.assert_allclose(res[1], ref, rtol=2e-5, atol=2e-5)


def test_presplit_needs_the_fp16_pair_kernel(lib):
  """a_presplit without Wh / a_amax is refused (no other kernel can read fp16 pairs)."""
  from epos_amd import _lib
  A = torch.zeros(64, 32, device='cuda')
  C = torch.zeros(64, 32, device='cuda')
  w = np.eye(32, dtype=np.float32)
  Wp = _pack(lib, w)
  a = _lib.PointwiseArgs(A=_p(A), lda=32, Wp=_p(Wp), bias=None, R=None, ldr=0, C=_p(C), ldc=32,
                         M=64, N=32, K=32, relu=0, relu_in=0, sub=1, a_presplit=1)
  with pytest.raises(_lib.EposError):
    _check(lib.epos_pointwise_conv_f32(ctypes.byref(a), None))


# ------------------------------- fused separable conv on the fp16-pair kernel (round 4) ---
def _sepconv_h2_problem(lib, b, h, w, cin, cout, rate, relu_in, relu_out, res, seed=0):
  """A separable conv with fp16-pair intermediates: (dw args, pw args, SepConvArgs) for given
  intermediate / output buffers; the depthwise input's absmax slot is measured."""
  from epos_amd import _lib
  rng = np.random.RandomState(seed + h * 7 + cin)
  x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
  w9c = (rng.standard_normal((9, cin)) / 3).astype(np.float32)
  dbias = rng.standard_normal(cin).astype(np.float32)
  wkn = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
  bias = rng.standard_normal(cout).astype(np.float32)
  npad = (cout + 127) // 128 * 128
  bpad = np.zeros(npad, np.float32); bpad[:cout] = bias
  m = b * h * w
  t = dict(x=x, w9c_h=w9c, dbias_h=dbias,
           X=torch.from_numpy(x).cuda(), w9c=torch.from_numpy(w9c).cuda(),
           dbias=torch.from_numpy(dbias).cuda(), Wp=_pack(lib, wkn), Wh=_pack(lib, wkn, 'h2'),
           bias=torch.from_numpy(bpad).cuda(),
           R=torch.from_numpy(rng.standard_normal((m, cout)).astype(np.float32)).cuda(),
           xs=_slot())
  _check(lib.epos_absmax_f32(_p(t['X']), cin, m, cin, _p(t['xs']), None))
  gain = float(np.abs(w9c.astype(np.float64)).sum(0).max())
  bias0 = float(np.abs(dbias).max())

  def make(T, C):
    dw = _lib.DepthwiseArgs(X=_p(t['X']), ldx=cin, w9c=_p(t['w9c']), bias=_p(t['dbias']),
                            Y=_p(T), ldy=cin, B=b, Hi=h, Wi=w, Ho=h, Wo=w, C=cin,
                            stride=1, rate=rate, relu_in=relu_in, relu_out=relu_out,
                            y_h2=1, x_amax=_p(t['xs']), gain=gain, bias0=bias0)
    pw = _lib.PointwiseArgs(A=_p(T), lda=cin, Wp=_p(t['Wp']), bias=_p(t['bias']),
                            R=_p(t['R']) if res else None, ldr=cout, C=_p(C), ldc=cout,
                            M=m, N=cout, K=cin, relu=1, relu_in=0, sub=1, Wh=_p(t['Wh']),
                            a_amax=_p(t['xs']), a_gain=gain, a_bias=bias0, a_presplit=1)
    return dw, pw, _lib.SepConvArgs(dw=dw, pw=pw)
  return t, m, make


@pytest.mark.parametrize('b,h,w,cin,cout,rate', [
    (1, 60, 80, 728, 728, 2),       # middle flow
    (2, 13, 17, 64, 128, 1),        # tiles straddle the two images
    (1, 9, 11, 36, 200, 3)])        # rate > a third of the image: every tap class at the border
@pytest.mark.parametrize('relu_in,relu_out,res', [(0, 1, 0), (1, 0, 1)])
def test_separable_conv_one_call_form(lib, b, h, w, cin, cout, rate, relu_in, relu_out, res):
  """epos_separable_conv_f32 (slim.separable_conv2d as one C-ABI call, fp16-pair intermediate)
  = epos_depthwise3x3_f32 (y_h2) + epos_pointwise_conv_f32 (a_presplit): intermediate and
  output bit for bit, and the intermediate against an fp64 depthwise conv. (Rounds 2 and 4
  ran it as ONE launch -- bit-identical, slower; since ABI 7 it issues the two launches.)"""
  from epos_amd import _lib
  t, m, make = _sepconv_h2_problem(lib, b, h, w, cin, cout, rate, relu_in, relu_out, res)
  assert t['Wh'] is not None
  T0 = torch.full((m, cin), 3.0, device='cuda'); C0 = torch.zeros(m, cout, device='cuda')
  dw, pw, _ = make(T0, C0)
  _check(lib.epos_depthwise3x3_f32(ctypes.byref(dw), None))
  _check(lib.epos_pointwise_conv_f32(ctypes.byref(pw), None))
  torch.cuda.synchronize()
  T1 = torch.full((m, cin), 7.0, device='cuda'); C1 = torch.full((m, cout), -1.0, device='cuda')
  _, _, sa = make(T1, C1)
  _check(lib.epos_separable_conv_f32(ctypes.byref(sa), None))
  torch.cuda.synchronize()
  assert torch.equal(T1.view(torch.int32), T0.view(torch.int32))
  assert torch.equal(C1, C0)
  bad = _lib.SepConvArgs(dw=dw, pw=pw)
  bad.pw.a_presplit = 0                  # the two halves must describe the same intermediate
  assert lib.epos_separable_conv_f32(ctypes.byref(bad), None) != 0
  # the reference itself against fp64 (so that "equal" means "right")
  x = torch.from_numpy(t['x']).double().permute(0, 3, 1, 2)
  if relu_in:
    x = x.clamp(min=0)
  wd = torch.from_numpy(t['w9c_h']).double().t().reshape(cin, 1, 3, 3)
  y = torch.nn.functional.conv2d(x, wd, padding=rate, dilation=rate, groups=cin) + \
      torch.from_numpy(t['dbias_h']).double().view(1, -1, 1, 1)
  if relu_out:
    y = y.clamp(min=0)
  y = y.permute(0, 2, 3, 1).reshape(m, cin).numpy()
  raw = T0.cpu().numpy().view(np.float16).reshape(m, cin // 4, 2, 4).astype(np.float64)
  bound = float(np.abs(t['w9c_h'].astype(np.float64)).sum(0).max()) * np.abs(t['x']).max() + \
      float(np.abs(t['dbias_h']).max())
  s = 2.0 ** (14 - np.floor(np.log2(bound)))
  dec = ((raw[:, :, 0, :] + raw[:, :, 1, :] / 2048.0) / s).reshape(m, cin)
  np.testing.assert_allclose(dec, y, rtol=1e-5, atol=1e-5 * bound)


# ------------------------------------------ softmax over 64-groups in the epilogue (round 4) ---


@pytest.mark.parametrize('b,hw,k,n,relu', [(1, 4800, 1536, 2048, 1), (2, 96, 64, 260, 0),
                                           (1, 6120, 128, 136, 1)])
def test_block_sums_in_the_epilogue_give_the_image_pooling_mean(lib, b, hw, k, n, relu):
  """EposPointwiseArgs.col_sums + epos_global_avg_pool_partial_f32 (round 4: the image-pooling
  mean of model.py:220 without re-reading the encoder output): the per-image channel means
  equal a float64 mean of the tensor the GEMM stored to a few fp32 roundings, the stored
  tensor itself is unchanged by the extra output, and two runs give the same bits."""
  from epos_amd import _lib
  rng = np.random.RandomState(hw + n)
  m = b * hw
  a = np.maximum(rng.standard_normal((m, k)), 0).astype(np.float32)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  A = torch.from_numpy(a).cuda()
  slot = _slot()
  _check(lib.epos_absmax_f32(_p(A), k, m, k, _p(slot), None))
  Wp, Wh = _pack(lib, w), _pack(lib, w, 'h2')
  bias = torch.from_numpy(np.pad(rng.standard_normal(n).astype(np.float32), (0, (-n) % 128))).cuda()
  blocks = (hw + 31) // 32
  if b > 1:
    assert hw % 32 == 0
  outs = []
  for with_sums in (0, 1, 1):
    C = torch.zeros(m, n, device='cuda')
    part = torch.full((b * blocks, n), 7.0, device='cuda')
    args = _lib.PointwiseArgs(A=_p(A), lda=k, Wp=_p(Wp), bias=_p(bias), R=None, ldr=0, C=_p(C),
                              ldc=n, M=m, N=n, K=k, relu=relu, relu_in=0, sub=1, Wh=_p(Wh),
                              a_amax=_p(slot), col_sums=_p(part) if with_sums else None, col_ld=n)
    _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
    Y = torch.zeros(b, n, device='cuda')
    if with_sums:
      _check(lib.epos_global_avg_pool_partial_f32(_p(part), n, _p(Y), b, blocks, n, hw, None))
    torch.cuda.synchronize()
    outs.append((C, Y))
  assert torch.equal(outs[0][0], outs[1][0])                    # C untouched by the extra output
  assert torch.equal(outs[1][1], outs[2][1])                    # deterministic
  ref = outs[1][0].cpu().numpy().astype(np.float64).reshape(b, hw, n).mean(1)
  got = outs[1][1].cpu().numpy().astype(np.float64)
  scale = np.abs(outs[1][0].cpu().numpy()).mean() + 1e-6
  assert np.abs(got - ref).max() <= 3e-6 * scale * np.sqrt(hw / 32.0) + 1e-7


def test_block_sums_need_the_fp16_pair_kernel(lib):
  from epos_amd import _lib
  A = torch.zeros(64, 32, device='cuda'); C = torch.zeros(64, 32, device='cuda')
  part = torch.zeros(2, 32, device='cuda')
  Wp = _pack(lib, np.eye(32, dtype=np.float32))
  a = _lib.PointwiseArgs(A=_p(A), lda=32, Wp=_p(Wp), bias=None, R=None, ldr=0, C=_p(C), ldc=32,
                         M=64, N=32, K=32, relu=0, relu_in=0, sub=1, col_sums=_p(part), col_ld=32)
  with pytest.raises(_lib.EposError):
    _check(lib.epos_pointwise_conv_f32(ctypes.byref(a), None))


def test_first_fp16_pair_launch_inside_a_capture_is_refused_not_broken(tmp_path):
  """The fp16-pair GEMM allocates its 64-byte zero chunk (and, without a caller-provided
  absmax slot, its slot ring) on the first launch of a device. Inside a stream capture
  that allocation / legacy-stream memset would invalidate the capture: the launch is REFUSED
  with an error there (a fresh process; the network plan warms up before it captures), the
  capture stays valid, and the same launch works afterwards -- also captured."""
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  script = r'''
import ctypes, sys, numpy as np, torch
sys.path.insert(0, %r)
from epos_amd import _lib
lib = _lib.load()
def p(t): return ctypes.c_void_p(t.data_ptr())
m, k, n = 256, 64, 128
rng = np.random.RandomState(0)
a = rng.standard_normal((m, k)).astype(np.float32); w = rng.standard_normal((k, n)).astype(np.float32)
A = torch.from_numpy(a).cuda(); C = torch.zeros(m, n, device='cuda')
def pack(fn, dt):            # the plain packer counts floats, the fp16-pair packer bytes
  tot = fn(w.ctypes.data_as(ctypes.c_void_p), k, n, None); d = np.empty(tot, dt)
  fn(w.ctypes.data_as(ctypes.c_void_p), k, n, d.ctypes.data_as(ctypes.c_void_p)); return torch.from_numpy(d).cuda()
Wh = pack(lib.epos_pack_pointwise_weights_h2, np.uint8); Wp = pack(lib.epos_pack_pointwise_weights, np.float32)
slot = torch.zeros(_lib.AMAX_WORDS, dtype=torch.int32, device='cuda')
_lib.check(lib.epos_absmax_f32(p(A), k, m, k, p(slot), None))
args = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Wp), bias=None, R=None, ldr=n, C=p(C), ldc=n, M=m, N=n, K=k,
                          relu=0, relu_in=0, sub=1, Wh=p(Wh), a_amax=p(slot))
torch.cuda.synchronize()
st = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
  rc = lib.epos_pointwise_conv_f32(ctypes.byref(args), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
msg = lib.epos_last_error().decode()
assert rc != 0 and 'capture' in msg, (rc, msg)
_lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))          # eager: allocates
torch.cuda.synchronize()
ref = a.astype(np.float64) @ w.astype(np.float64)
assert np.abs(C.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()
C.zero_()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=st):
  _lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(args), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
g2.replay(); torch.cuda.synchronize()
assert np.abs(C.cpu().numpy() - ref).max() < 1e-4 * np.abs(ref).max()
print('OK')
''' % root
  r = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and 'OK' in r.stdout, r.stdout + r.stderr
