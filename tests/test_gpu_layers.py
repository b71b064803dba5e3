"""GPU parity tests of the layer kernels, called through the C ABI, against the
torch-CPU oracle (oracle/net_ref.py) on identical seeded inputs. fp32 tolerance:
rtol = atol = 1e-4 (the bar slim's own atrous tests use, resnet_v1_test.py:239),
tightened where the arithmetic is order-identical."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


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


def _p(t, off=0):
  return ctypes.c_void_p(t.data_ptr() + off * t.element_size())


def _pack(lib, w_kn):
  k, n = w_kn.shape
  total = lib.epos_pack_pointwise_weights(None, k, n, None)
  dst = np.empty(total, np.float32)
  w = np.ascontiguousarray(w_kn, np.float32)
  lib.epos_pack_pointwise_weights(w.ctypes.data_as(ctypes.c_void_p), k, n,
                                  dst.ctypes.data_as(ctypes.c_void_p))
  return torch.from_numpy(dst).cuda()


@pytest.mark.parametrize('m,k,n', [(300, 64, 128), (4800, 728, 728), (77, 28, 32),
                                   (1000, 304, 256), (513, 256, 22), (1, 2048, 256),
                                   (256, 1280, 256), (130, 36, 1344)])
@pytest.mark.parametrize('relu,relu_in,res', [(0, 0, 0), (1, 0, 1), (0, 1, 0)])
def test_pointwise_gemm(lib, m, k, n, relu, relu_in, res):
  from epos_amd import _lib
  rng = np.random.RandomState(m + k + n)
  a = rng.standard_normal((m, k)).astype(np.float32)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  bias = rng.standard_normal(n).astype(np.float32)
  r = rng.standard_normal((m, n)).astype(np.float32)
  npad = (n + 127) // 128 * 128
  bpad = np.zeros(npad, np.float32); bpad[:n] = bias
  A, Wp, Bd, R = (torch.from_numpy(a).cuda(), _pack(lib, w),
                  torch.from_numpy(bpad).cuda(), torch.from_numpy(r).cuda())
  ldc = n + 5                      # write into a wider buffer at an offset
  C = torch.full((m, ldc), -7.0, device='cuda')
  args = _lib.PointwiseArgs(A=_p(A), lda=k, Wp=_p(Wp), bias=_p(Bd),
                            R=_p(R) if res else None, ldr=n, C=_p(C, 3), ldc=ldc,
                            M=m, N=n, K=k, relu=relu, relu_in=relu_in, sub=1)
  _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
  torch.cuda.synchronize()
  out = C.cpu().numpy()
  aa = np.maximum(a, 0) if relu_in else a
  ref = aa.astype(np.float64) @ w.astype(np.float64) + bias
  if res:
    ref = ref + r
  if relu:
    ref = np.maximum(ref, 0)
  np.testing.assert_allclose(out[:, 3:3 + n], ref, rtol=1e-4, atol=1e-4)
  assert (out[:, :3] == -7.0).all() and (out[:, 3 + n:] == -7.0).all()


@pytest.mark.parametrize('k', [32, 40, 64, 92, 96, 128, 132, 160, 224, 736])
@pytest.mark.parametrize('m,n,aligned,res', [(64, 128, 1, 0), (130, 200, 1, 1),
                                             (257, 132, 0, 1), (1000, 96, 1, 0)])
def test_pointwise_gemm_dma_ring(lib, k, m, n, aligned, res):
  """LDS-DMA kernel (the default for N > 64): every prologue / steady / tail path
  of the three-stage ring (1..23 K tiles, partial last K tile with and without
  whole k-groups missing), ragged M and N tiles, the float4 epilogue (16-byte
  aligned rows) and the scalar one, residual + ReLU. A is followed by NaNs so that
  reading past K inside the last K tile would poison the result."""
  from epos_amd import _lib
  rng = np.random.RandomState(k * 7 + m + n)
  a = rng.standard_normal((m, k)).astype(np.float32)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  bias = rng.standard_normal(n).astype(np.float32)
  r = rng.standard_normal((m, n)).astype(np.float32)
  npad = (n + 127) // 128 * 128
  bpad = np.zeros(npad, np.float32); bpad[:n] = bias
  lda = k + 36                                 # columns k.. of every row are NaN
  abuf = np.full((m, lda), np.nan, np.float32); abuf[:, :k] = a
  A, Wp, Bd, R = (torch.from_numpy(abuf).cuda(), _pack(lib, w),
                  torch.from_numpy(bpad).cuda(), torch.from_numpy(r).cuda())
  ldc = n + (4 if aligned else 5)
  off = 4 if aligned else 3
  C = torch.full((m, ldc), -7.0, device='cuda')
  args = _lib.PointwiseArgs(A=_p(A), lda=lda, Wp=_p(Wp), bias=_p(Bd),
                            R=_p(R) if res else None, ldr=n, C=_p(C, off), ldc=ldc,
                            M=m, N=n, K=k, relu=res, relu_in=0, sub=1)
  _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
  torch.cuda.synchronize()
  out = C.cpu().numpy()
  ref = a.astype(np.float64) @ w.astype(np.float64) + bias
  if res:
    ref = np.maximum(ref + r, 0)
  np.testing.assert_allclose(out[:, off:off + n], ref, rtol=1e-4, atol=1e-4)
  assert (out[:, :off] == -7.0).all() and (out[:, off + n:] == -7.0).all()


def _pack_split(lib, w_kn):
  k, n = w_kn.shape
  total = lib.epos_pack_pointwise_weights_split(None, k, n, None)
  dst = np.empty(total, np.uint8)
  w = np.ascontiguousarray(w_kn, np.float32)
  lib.epos_pack_pointwise_weights_split(w.ctypes.data_as(ctypes.c_void_p), k, n,
                                        dst.ctypes.data_as(ctypes.c_void_p))
  return torch.from_numpy(dst).cuda()


@pytest.mark.parametrize('k', [16, 32, 48, 64, 80, 96, 112, 128, 144, 40, 92, 728])
@pytest.mark.parametrize('m,n,aligned,res', [(128, 128, 1, 0), (130, 200, 1, 1),
                                             (257, 132, 0, 1), (1000, 96, 1, 0),
                                             (16700, 500, 1, 1), (16450, 490, 0, 0)])
def test_pointwise_gemm_split_ring(lib, k, m, n, aligned, res):
  """Split-operand kernel (fp32 GEMM on the bf16 matrix pipe, taken when Ws is given):
  every prologue / steady / tail path of the four-stage ring (1..46 K steps, partial
  last step) in both tile shapes (64 x 128 for small grids, 128 x 128 from 512 tiles
  on: the two 16 000-row cases), ragged M and N tiles, the float4 and the scalar epilogue, residual +
  ReLU; NaNs behind every row of A poison any read past K. Tolerance as for the fp32
  kernels (the accuracy comparison proper is test_pointwise_gemm_split_accuracy)."""
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
  A, Wp, Ws, Bd, R = (torch.from_numpy(abuf).cuda(), _pack(lib, w), _pack_split(lib, w),
                      torch.from_numpy(bpad).cuda(), torch.from_numpy(r).cuda())
  ldc = n + (4 if aligned else 5)
  off = 4 if aligned else 3
  C = torch.full((m, ldc), -7.0, device='cuda')
  args = _lib.PointwiseArgs(A=_p(A), lda=lda, Wp=_p(Wp), bias=_p(Bd),
                            R=_p(R) if res else None, ldr=n, C=_p(C, off), ldc=ldc,
                            M=m, N=n, K=k, relu=res, relu_in=0, sub=1, Ws=_p(Ws))
  _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
  torch.cuda.synchronize()
  out = C.cpu().numpy()
  ref = a.astype(np.float64) @ w.astype(np.float64) + bias
  if res:
    ref = np.maximum(ref + r, 0)
  np.testing.assert_allclose(out[:, off:off + n], ref, rtol=2e-5, atol=2e-5)
  assert (out[:, :off] == -7.0).all() and (out[:, off + n:] == -7.0).all()


def test_pointwise_gemm_split_grouped_and_strided(lib):
  """Three problems in one grid (one of them 22 columns wide: scalar epilogue inside
  a group) and a stride-2 row gather, through the split kernel."""
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
    Wp, Ws = _pack(lib, w), _pack_split(lib, w)
    C = torch.zeros(b * ho * wo, n, device='cuda')
    keep += [Wp, Ws, C]
    arr[i] = _lib.PointwiseArgs(A=_p(X), lda=cin, Wp=_p(Wp), bias=None, R=None, ldr=0,
                                C=_p(C), ldc=n, M=b * ho * wo, N=n, K=cin, relu=0,
                                relu_in=0, sub=sub, Ho=ho, Wo=wo, Hi=hi, Wi=wi,
                                Ws=_p(Ws))
    outs.append(C)
    refs.append(x[:, ::2, ::2, :].reshape(-1, cin).astype(np.float64) @ w)
  _check(lib.epos_pointwise_conv_grouped_f32(arr, 3, None))
  torch.cuda.synchronize()
  for C, ref in zip(outs, refs):
    np.testing.assert_allclose(C.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('m,k,n', [(4800, 728, 728), (2048, 2048, 256), (4096, 256, 1344)])
def test_pointwise_gemm_split_accuracy(lib, m, k, n):
  """The claim the split kernel rests on: its error against an fp64 product is NOT
  larger than the fp32-MFMA kernel's on the same inputs (ReLU-like activations, as in
  the network). Errors are measured relative to sum_k |a||w| per output."""
  from epos_amd import _lib
  rng = np.random.RandomState(k + n)
  a = np.maximum(rng.standard_normal((m, k)), 0).astype(np.float32)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  A, Wp, Ws = torch.from_numpy(a).cuda(), _pack(lib, w), _pack_split(lib, w)
  ref = a.astype(np.float64) @ w.astype(np.float64)
  mag = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64)
  errs = {}
  for name, ws in (('fp32_mfma', None), ('split', Ws)):
    C = torch.zeros(m, n, device='cuda')
    args = _lib.PointwiseArgs(A=_p(A), lda=k, Wp=_p(Wp), bias=None, R=None, ldr=0,
                              C=_p(C), ldc=n, M=m, N=n, K=k, relu=0, relu_in=0, sub=1,
                              Ws=_p(ws) if ws is not None else None)
    _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
    torch.cuda.synchronize()
    e = np.abs(C.cpu().numpy().astype(np.float64) - ref) / mag
    errs[name] = (float(np.sqrt((e * e).mean())), float(e.max()))
  print('rms / max error relative to sum|a||w|:', errs)
  assert errs['split'][0] <= errs['fp32_mfma'][0] * 1.05
  assert errs['split'][1] <= errs['fp32_mfma'][1] * 1.5
  assert errs['split'][1] < 4e-7


def test_pointwise_gemm_dma_grouped_and_strided(lib):
  """One grid for three problems of different shapes (as the heads / ASPP groups)
  plus a stride-2 row gather (shortcut convs), all through the LDS-DMA kernel."""
  from epos_amd import _lib
  rng = np.random.RandomState(11)
  b, hi, wi, cin = 2, 13, 18, 72
  ho, wo = (hi + 1) // 2, (wi + 1) // 2
  x = rng.standard_normal((b, hi, wi, cin)).astype(np.float32)
  X = torch.from_numpy(x).cuda()
  outs, refs, arr = [], [], (_lib.PointwiseArgs * 3)()
  keep = []
  for i, (n, sub) in enumerate([(136, 2), (80, 2), (260, 2)]):
    w = (rng.standard_normal((cin, n)) / np.sqrt(cin)).astype(np.float32)
    Wp = _pack(lib, w)
    C = torch.zeros(b * ho * wo, n, device='cuda')
    keep += [Wp, C]
    arr[i] = _lib.PointwiseArgs(A=_p(X), lda=cin, Wp=_p(Wp), bias=None, R=None, ldr=0,
                                C=_p(C), ldc=n, M=b * ho * wo, N=n, K=cin, relu=0,
                                relu_in=0, sub=sub, Ho=ho, Wo=wo, Hi=hi, Wi=wi)
    outs.append(C)
    refs.append(x[:, ::2, ::2, :].reshape(-1, cin).astype(np.float64) @ w)
  _check(lib.epos_pointwise_conv_grouped_f32(arr, 3, None))
  torch.cuda.synchronize()
  for C, ref in zip(outs, refs):
    np.testing.assert_allclose(C.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


def test_pointwise_gemm_layout_is_not_transposed(lib):
  """A = I with an ASYMMETRIC W must reproduce W (catches row/col swaps)."""
  from epos_amd import _lib
  k = n = 64
  w = np.arange(k * n, dtype=np.float32).reshape(k, n)
  A = torch.eye(k, device='cuda')
  C = torch.zeros(k, n, device='cuda')
  Wp = _pack(lib, w)
  args = _lib.PointwiseArgs(A=_p(A), lda=k, Wp=_p(Wp), bias=None, R=None, ldr=0,
                            C=_p(C), ldc=n, M=k, N=n, K=k, relu=0, relu_in=0,
                            sub=1)
  _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
  assert np.array_equal(C.cpu().numpy(), w)


def test_pointwise_stride2_shortcut(lib):
  """1x1 stride-2 'SAME' conv (net_xception.py:296-302) samples even pixels."""
  from epos_amd import _lib
  from oracle import net_ref
  rng = np.random.RandomState(0)
  b, hi, wi, cin, cout = 2, 9, 12, 16, 40
  x = rng.standard_normal((b, hi, wi, cin)).astype(np.float32)
  w = rng.standard_normal((1, 1, cin, cout)).astype(np.float32)
  ref = net_ref.conv2d_raw(torch.from_numpy(x).permute(0, 3, 1, 2), w, 2, 1,
                           'SAME').permute(0, 2, 3, 1).numpy()
  ho, wo = ref.shape[1], ref.shape[2]
  X = torch.from_numpy(x).cuda()
  C = torch.zeros(b, ho, wo, cout, device='cuda')
  Wp = _pack(lib, w.reshape(cin, cout))
  args = _lib.PointwiseArgs(A=_p(X), lda=cin, Wp=_p(Wp), bias=None, R=None,
                            ldr=0, C=_p(C), ldc=cout, M=b * ho * wo, N=cout,
                            K=cin, relu=0, relu_in=0, sub=2, Ho=ho, Wo=wo, Hi=hi,
                            Wi=wi)
  _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
  np.testing.assert_allclose(C.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('b,h,w,cin,cout,stride,rate', [
    (2, 12, 16, 32, 64, 1, 1), (1, 9, 21, 64, 40, 1, 1), (2, 7, 5, 32, 136, 1, 1),
    (1, 30, 44, 96, 64, 1, 1), (2, 15, 21, 64, 72, 2, 1), (1, 16, 24, 32, 64, 2, 1),
    (1, 20, 28, 64, 130, 1, 2), (2, 13, 17, 32, 48, 1, 4), (4, 60, 80, 32, 64, 1, 1)])
@pytest.mark.parametrize('split', [0, 1])
def test_conv3x3_implicit_gemm(lib, b, h, w, cin, cout, stride, rate, split):
  """Dense 3x3 'SAME' conv as an implicit GEMM (taps gathered by the LDS-DMA, zero
  block outside the image) vs conv2d_same of the oracle
  (external/slim/nets/resnet_utils.py:77-122), through the fp32-MFMA kernel (both
  tile layouts: Cout <= 64 and > 64) and, with the split-packed weights, through the
  split-operand kernel (both tile shapes: the last case has >= 200 tiles); one and
  several channel blocks per tap, ragged M tiles."""
  from epos_amd import _lib
  from oracle import net_ref
  rng = np.random.RandomState(b * h + cin)
  x = rng.standard_normal((b, h, w, cin)).astype(np.float32)
  wgt = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
  bias = rng.standard_normal(cout).astype(np.float32)
  xt = torch.from_numpy(x).permute(0, 3, 1, 2)
  if stride == 1:
    ref = net_ref.conv2d_raw(xt, wgt, 1, rate, 'SAME')
  else:                               # conv2d_same: explicit pad + VALID
    ref = net_ref.conv2d_raw(net_ref.fixed_padding(xt, 3, rate), wgt, stride, rate, 'VALID')
  ref = np.maximum(ref.permute(0, 2, 3, 1).numpy() + bias, 0)
  ho, wo = ref.shape[1], ref.shape[2]
  X = torch.from_numpy(x).cuda()
  Wp = _pack(lib, wgt.reshape(9 * cin, cout))
  npad = (cout + 127) // 128 * 128
  bpad = np.zeros(npad, np.float32); bpad[:cout] = bias
  Bd = torch.from_numpy(bpad).cuda()
  Y = torch.full((b, ho, wo, cout), -3.0, device='cuda')
  Ws = _pack_split(lib, wgt.reshape(9 * cin, cout)) if split else None
  args = _lib.Conv3x3Args(X=_p(X), ldx=cin, Wp=_p(Wp), bias=_p(Bd), Y=_p(Y), ldy=cout,
                          B=b, H=h, W=w, Cin=cin, Cout=cout, stride=stride, rate=rate,
                          relu=1, Ws=_p(Ws) if split else None)
  _check(lib.epos_conv3x3_f32(ctypes.byref(args), None))
  torch.cuda.synchronize()
  np.testing.assert_allclose(Y.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('hi,wi,c,stride,rate', [
    (17, 23, 32, 1, 1), (16, 24, 64, 2, 1), (15, 21, 8, 2, 1), (20, 28, 16, 1, 2),
    (60, 80, 8, 1, 12), (30, 40, 8, 1, 36), (9, 9, 12, 1, 4)])
@pytest.mark.parametrize('relu_in,relu_out', [(0, 0), (1, 0), (0, 1)])
def test_depthwise(lib, hi, wi, c, stride, rate, relu_in, relu_out):
  from epos_amd import _lib
  from oracle import net_ref
  rng = np.random.RandomState(hi * wi + c)
  b = 2
  x = rng.standard_normal((b, hi, wi, c)).astype(np.float32)
  w = rng.standard_normal((3, 3, c, 1)).astype(np.float32)
  scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
  bias = rng.standard_normal(c).astype(np.float32)
  xt = torch.from_numpy(x).permute(0, 3, 1, 2)
  if relu_in:
    xt = F.relu(xt)
  if stride == 1:
    y = net_ref.depthwise_raw(xt, w, 1, rate, 'SAME')
  else:
    y = net_ref.depthwise_raw(net_ref.fixed_padding(xt, 3, rate), w, stride, rate,
                              'VALID')
  y = y * torch.from_numpy(scale).view(1, -1, 1, 1) + torch.from_numpy(bias).view(
      1, -1, 1, 1)
  if relu_out:
    y = F.relu(y)
  ref = y.permute(0, 2, 3, 1).numpy()
  ho, wo = ref.shape[1], ref.shape[2]
  w9c = (w[:, :, :, 0].reshape(9, c) * scale[None]).astype(np.float32)
  X, Wd, Bd = (torch.from_numpy(x).cuda(), torch.from_numpy(w9c).cuda(),
               torch.from_numpy(bias).cuda())
  Y = torch.zeros(b, ho, wo, c, device='cuda')
  args = _lib.DepthwiseArgs(X=_p(X), ldx=c, w9c=_p(Wd), bias=_p(Bd), Y=_p(Y),
                            ldy=c, B=b, Hi=hi, Wi=wi, Ho=ho, Wo=wo, C=c,
                            stride=stride, rate=rate, relu_in=relu_in,
                            relu_out=relu_out)
  _check(lib.epos_depthwise3x3_f32(ctypes.byref(args), None))
  np.testing.assert_allclose(Y.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('hi,wi,c,rate,h2', [(60, 80, 728, 2, 0), (60, 80, 728, 2, 1),
                                             (17, 23, 728, 1, 1), (30, 40, 1024, 2, 1),
                                             (24, 20, 264, 1, 0)])
def test_depthwise_line_aligned_rows_equal_dense_rows(lib, hi, wi, c, rate, h2):
  """Rows padded to a multiple of 32 floats (what the plan allocates since round 6: 728 -> 736)
  take the line-aligned channel slices (an XCD's slice starts on a 128-byte line); the values
  are those of the dense-row launch bit for bit, fp32 and fp16-pair output alike, and the
  padding columns are never written."""
  from epos_amd import _lib
  rng = np.random.RandomState(c + hi)
  b = 2
  ld = (c + 31) // 32 * 32
  assert ld != c or c % 32 == 0
  x = rng.standard_normal((b, hi, wi, c)).astype(np.float32)
  w9c = rng.standard_normal((9, c)).astype(np.float32)
  bias = rng.standard_normal(c).astype(np.float32)
  Wd, Bd = torch.from_numpy(w9c).cuda(), torch.from_numpy(bias).cuda()
  slot = torch.zeros(64, dtype=torch.int32, device='cuda')
  slot[5] = int(np.float32(np.abs(x).max()).view(np.int32))
  gain = float(np.abs(w9c.astype(np.float64)).sum(0).max())
  outs = []
  for pad in (False, True):
    l = ld if pad else c
    X = torch.full((b, hi, wi, l), 7.0, device='cuda')
    X[..., :c] = torch.from_numpy(x).cuda()
    Y = torch.full((b, hi, wi, l), -3.0, device='cuda')
    args = _lib.DepthwiseArgs(X=_p(X), ldx=l, w9c=_p(Wd), bias=_p(Bd), Y=_p(Y), ldy=l, B=b,
                              Hi=hi, Wi=wi, Ho=hi, Wo=wi, C=c, stride=1, rate=rate, relu_in=1,
                              relu_out=0)
    if h2:
      args.y_h2 = 1
      args.x_amax = _p(slot)
      args.gain, args.bias0 = gain, float(np.abs(bias).max())
    _check(lib.epos_depthwise3x3_f32(ctypes.byref(args), None))
    torch.cuda.synchronize()
    out = Y.cpu().numpy()
    if pad and l != c:
      assert (out[..., c:] == -3.0).all()
    outs.append(out[..., :c].copy())
  assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


@pytest.mark.parametrize('hi,wi,cin,cout,stride,pre', [(16, 20, 3, 32, 2, 1),
                                                       (15, 21, 3, 8, 2, 1),
                                                       (12, 16, 32, 64, 1, 0)])
def test_stem_conv_im2col_gemm(lib, hi, wi, cin, cout, stride, pre):
  """conv2d_same (resnet_utils.py:77-122) = im2col + GEMM; includes the slim KAT
  geometry (explicit pad 1 + VALID for stride 2)."""
  from epos_amd import _lib
  from oracle import net_ref
  rng = np.random.RandomState(1)
  b = 2
  x = (rng.uniform(0, 255, (b, hi, wi, cin)) if pre else
       rng.standard_normal((b, hi, wi, cin))).astype(np.float32)
  w = rng.standard_normal((3, 3, cin, cout)).astype(np.float32) * 0.1
  xt = torch.from_numpy(x).permute(0, 3, 1, 2)
  if pre:
    xt = (2.0 / 255.0) * xt - 1.0
  ref = net_ref.conv2d_same_raw(xt, w, stride).permute(0, 2, 3, 1).numpy()
  ho, wo = ref.shape[1], ref.shape[2]
  k = 9 * cin
  ld = (k + 3) // 4 * 4
  X = torch.from_numpy(x).cuda()
  col = torch.full((b * ho * wo, ld), 9.0, device='cuda')
  ia = _lib.Im2colArgs(X=_p(X), ldx=cin, col=_p(col), ldcol=ld, B=b, Hi=hi, Wi=wi,
                       Ho=ho, Wo=wo, C=cin, stride=stride, rate=1, pad=1,
                       preprocess=pre)
  _check(lib.epos_im2col3x3_f32(ctypes.byref(ia), None))
  wk = np.zeros((ld, cout), np.float32); wk[:k] = w.reshape(k, cout)
  C = torch.zeros(b * ho * wo, cout, device='cuda')
  args = _lib.PointwiseArgs(A=_p(col), lda=ld, Wp=_p(_pack(lib, wk)), bias=None,
                            R=None, ldr=0, C=_p(C), ldc=cout, M=b * ho * wo,
                            N=cout, K=ld, relu=0, relu_in=0, sub=1)
  _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
  np.testing.assert_allclose(C.cpu().numpy().reshape(ref.shape), ref, rtol=1e-4,
                             atol=1e-4)


def test_slim_conv2d_same_kat_on_device(lib):
  """external/slim/nets/resnet_v1_test.py:72-109 on the HIP path: x,w = i+j grids,
  4x4 input, stride 2 -> [[14,43],[43,84]] (NOT TF-SAME's [[48,37],[37,22]])."""
  from epos_amd import _lib
  x = np.add.outer(np.arange(4), np.arange(4)).astype(np.float32).reshape(1, 4, 4, 1)
  w = np.add.outer(np.arange(3), np.arange(3)).astype(np.float32).reshape(9, 1)
  X = torch.from_numpy(x).cuda()
  col = torch.zeros(4, 12, device='cuda')
  ia = _lib.Im2colArgs(X=_p(X), ldx=1, col=_p(col), ldcol=12, B=1, Hi=4, Wi=4,
                       Ho=2, Wo=2, C=1, stride=2, rate=1, pad=1, preprocess=0)
  _check(lib.epos_im2col3x3_f32(ctypes.byref(ia), None))
  wk = np.zeros((12, 1), np.float32); wk[:9] = w
  C = torch.zeros(4, 1, device='cuda')
  args = _lib.PointwiseArgs(A=_p(col), lda=12, Wp=_p(_pack(lib, wk)), bias=None,
                            R=None, ldr=0, C=_p(C), ldc=1, M=4, N=1, K=12, relu=0,
                            relu_in=0, sub=1)
  _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
  assert C.cpu().numpy().reshape(2, 2).tolist() == [[14, 43], [43, 84]]


def test_global_avg_pool_and_resize(lib):
  from epos_amd import _lib
  from oracle import net_ref
  rng = np.random.RandomState(3)
  b, h, w, c = 2, 15, 20, 72
  x = rng.standard_normal((b, h, w, c)).astype(np.float32)
  X = torch.from_numpy(x).cuda()
  Y = torch.zeros(b, c, device='cuda')
  _check(lib.epos_global_avg_pool_f32(_p(X), c, _p(Y), b, h * w, c, None))
  np.testing.assert_allclose(Y.cpu().numpy(), x.mean(axis=(1, 2)), rtol=1e-5,
                             atol=1e-6)
  for (ho, wo) in [(29, 39), (30, 40), (h, w)]:
    ref = net_ref.resize_bilinear_align_corners(
        torch.from_numpy(x).permute(0, 3, 1, 2), (ho, wo)).permute(0, 2, 3, 1).numpy()
    Z = torch.zeros(b, ho, wo, c + 8, device='cuda')
    _check(lib.epos_resize_bilinear_f32(_p(X), c, _p(Z, 4), c + 8, b, h, w, ho,
                                            wo, c, None))
    np.testing.assert_allclose(Z.cpu().numpy()[..., 4:4 + c], ref, rtol=1e-5,
                               atol=1e-5)
  # broadcast from 1x1 (image pooling branch)
  P1 = torch.from_numpy(x[:, :1, :1, :].copy()).cuda()
  Z = torch.zeros(b, 6, 7, c, device='cuda')
  _check(lib.epos_resize_bilinear_f32(_p(P1), c, _p(Z), c, b, 1, 1, 6, 7, c,
                                          None))
  assert np.array_equal(Z.cpu().numpy(), np.broadcast_to(x[:, :1, :1, :],
                                                         (b, 6, 7, c)))


@pytest.mark.parametrize('g', [2, 22, 31, 64])
def test_softmax_and_argmax(lib, g):
  from epos_amd import _lib
  rng = np.random.RandomState(g)
  n = 1000
  x = (rng.standard_normal((n, g)) * 3).astype(np.float32)
  x[5, :] = 1.25                                    # exact tie -> first index
  X = torch.from_numpy(x).cuda()
  lab = torch.zeros(n, dtype=torch.int64, device='cuda')
  _check(lib.epos_softmax_groups_f32(_p(X), n, g, None))
  _check(lib.epos_argmax_i64(_p(X), g, _p(lab), n, g, None))
  ref = torch.softmax(torch.from_numpy(x), dim=-1).numpy()
  out = X.cpu().numpy()
  np.testing.assert_allclose(out, ref, rtol=2e-6, atol=1e-7)
  assert np.array_equal(lab.cpu().numpy(), out.argmax(axis=1))
  assert lab[5].item() == 0


def test_clock_probe_reports_a_plausible_core_clock(lib):
  """epos_clock_probe: shader cycles per 100 MHz tick while a wave spins for 200 us."""
  from epos_amd import _lib
  out = torch.zeros(2, dtype=torch.int64, device='cuda')
  _check(lib.epos_clock_probe(_p(out), 200, None))
  torch.cuda.synchronize()
  cyc, ticks = [int(v) for v in out.cpu()]
  assert 19000 <= ticks <= 40000                    # ~200 us of the 100 MHz counter
  mhz = cyc / ticks * 100.0
  assert 500.0 < mhz < 3000.0, mhz


@pytest.mark.parametrize('env', [{'EPOS_GEMM_SPLIT': '0'}, {'EPOS_GEMM_SPLIT_ACC': '1'},
                                 {'EPOS_GEMM_SPLIT_ROWS': '64'},
                                 {'EPOS_GEMM_SPLIT_ROWS': '128'}])
def test_pointwise_gemm_split_switches(env):
  """The process-wide switches of the split-operand GEMM (read once per process, hence a
  subprocess each): fp32-MFMA kernel instead, one accumulator, forced tile height -- all
  must give the same result within the fp32 tolerance."""
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
rng = np.random.RandomState(5)
for (m, k, n) in [(700, 728, 200), (16500, 80, 520)]:
  a = np.maximum(rng.standard_normal((m, k)), 0).astype(np.float32)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  tot = lib.epos_pack_pointwise_weights(None, k, n, None); d = np.empty(tot, np.float32)
  lib.epos_pack_pointwise_weights(w.ctypes.data_as(ctypes.c_void_p), k, n, d.ctypes.data_as(ctypes.c_void_p))
  tot = lib.epos_pack_pointwise_weights_split(None, k, n, None); d8 = np.empty(tot, np.uint8)
  lib.epos_pack_pointwise_weights_split(w.ctypes.data_as(ctypes.c_void_p), k, n, d8.ctypes.data_as(ctypes.c_void_p))
  A, Wp, Ws = torch.from_numpy(a).cuda(), torch.from_numpy(d).cuda(), torch.from_numpy(d8).cuda()
  C = torch.zeros(m, n, device='cuda')
  args = _lib.PointwiseArgs(A=p(A), lda=k, Wp=p(Wp), bias=None, R=None, ldr=0, C=p(C), ldc=n,
                            M=m, N=n, K=k, relu=0, relu_in=0, sub=1, Ws=p(Ws))
  _lib.check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
  torch.cuda.synchronize()
  ref = a.astype(np.float64) @ w.astype(np.float64)
  np.testing.assert_allclose(C.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
print('ok')
''' % root
  if env.get('EPOS_GEMM_SPLIT') == '0':       # the fp32-MFMA kernels live in the test build
    from epos_amd import build
    env = dict(env, EPOS_HIP_LIB=build.REF_LIB_PATH)
  r = subprocess.run([sys.executable, '-c', script], env=dict(os.environ, **env),
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and 'ok' in r.stdout, r.stdout + r.stderr


# ------------------------------------------------------- fused separable conv ---


# --------------------------------------------- split GEMM on adversarial operands ---
def _split_vs_fp32(lib, a, w, split):
  """C = a @ w through epos_pointwise_conv_f32, with (split kernel) or without (fp32-MFMA
  kernel) the split-packed weights."""
  from epos_amd import _lib
  m, k = a.shape
  n = w.shape[1]
  A = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
  C = torch.zeros(m, n, device='cuda')
  Wp = _pack(lib, w)
  Ws = _pack_split(lib, w) if split else None
  args = _lib.PointwiseArgs(A=_p(A), lda=k, Wp=_p(Wp), bias=None, R=None, ldr=n, C=_p(C),
                            ldc=n, M=m, N=n, K=k, relu=0, relu_in=0, sub=1,
                            Ws=_p(Ws) if split else None)
  _check(lib.epos_pointwise_conv_f32(ctypes.byref(args), None))
  torch.cuda.synchronize()
  return C.cpu().numpy().astype(np.float64)


def test_split_gemm_adversarial_operands(lib):
  """The fp32-equivalence claim of the split-operand GEMM outside the comfortable range
  (network activations are O(1e-3 .. 1e3)): explicit bounds against an fp64 product, next
  to the fp32-MFMA kernel on the same operands. eps = 2^-24 (fp32 unit round-off); errors
  are measured relative to sum_k |a_k| |w_k|, the natural scale of a dot product."""
  rng = np.random.RandomState(0)
  eps = 2.0 ** -24
  m, k, n = 256, 512, 128

  rms = {}

  def rel_err(a, w):
    ref = a.astype(np.float64) @ w.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64)
    out = {}
    for split in (1, 0):
      c = _split_vs_fp32(lib, a, w, split)
      with np.errstate(invalid='ignore', divide='ignore'):
        r = np.abs(c - ref) / np.maximum(scale, 1e-300)
      out[split] = np.nanmax(r)
      rms[split] = float(np.sqrt(np.nanmean(r * r)))
    return out

  # 1. alternating-sign cancellation: the exact result is tiny against the terms
  v = rng.uniform(1, 2, (m, k)).astype(np.float32)
  a = v * np.where(np.arange(k) % 2 == 0, 1.0, -1.0).astype(np.float32)
  a[:, 1::2] = -a[:, 0::2] * (1 + rng.uniform(-1e-6, 1e-6, (m, k // 2)).astype(np.float32))
  w = np.ones((k, n), np.float32) * rng.uniform(0.5, 1.5, (1, n)).astype(np.float32)
  e = rel_err(a, w)
  assert e[1] <= 2 * eps and e[1] <= e[0] * 1.01 + eps / 8, e
  # 2. magnitudes spread over 2^-60 .. 2^60 in BOTH operands inside one dot product
  #    (products from 2^-120 to 2^120)
  a = (rng.uniform(1, 2, (m, k)) * 2.0 ** rng.randint(-60, 61, (m, k)) *
       rng.choice([-1, 1], (m, k))).astype(np.float32)
  w = (rng.uniform(1, 2, (k, n)) * 2.0 ** rng.randint(-60, 61, (k, n)) *
       rng.choice([-1, 1], (k, n))).astype(np.float32)
  e = rel_err(a, w)
  # one huge term dominates sum |a||w| here and every later addition rounds at ITS
  # magnitude: an fp32 chain is at ~1 eps rms / ~10 eps max on such data (measured: the
  # fp32-MFMA kernel 1.1-1.8 eps rms, 8-15 max), and so is the split kernel -- its
  # advantage (0.04 vs 0.31 eps rms at equal magnitudes) comes from the products, not
  # from the order of the additions. Bound: same class as the fp32 kernel.
  assert e[1] <= 16 * eps and rms[1] <= 1.25 * rms[0] + eps / 8, (e, rms)
  # 3. small magnitudes whose low bf16 piece is a DENORMAL (|x| ~ 2^-118 .. 2^-108: x itself
  #    is a normal fp32 number, lo = x - hi - mid lies below 2^-126). If the matrix pipe
  #    flushed denormal bf16 inputs the third-order terms would be lost and the error
  #    would jump to ~2^-16; measured: it does not (9 eps max here, the fp32-MFMA kernel 25
  #    on the same operands, whose magnitudes span 11 octaves -- see case 2).
  a = (rng.uniform(1, 2, (m, k)) * 2.0 ** rng.randint(-118, -107, (m, k))).astype(np.float32)
  w = rng.uniform(1, 2, (k, n)).astype(np.float32)
  e = rel_err(a, w)
  assert e[1] <= 32 * eps and rms[1] <= 1.25 * rms[0] + eps / 8, (e, rms)
  # 4. non-finite operands stay non-finite (an Inf splits into Inf + NaN pieces: the row
  #    comes out NaN where an fp32 chain would give +-Inf); finite rows are untouched
  a = rng.standard_normal((m, k)).astype(np.float32)
  w = rng.standard_normal((k, n)).astype(np.float32)
  a[3, 17] = np.inf
  a[5, 100] = np.nan
  c = _split_vs_fp32(lib, a, w, 1)
  assert not np.isfinite(c[3]).any() and np.isnan(c[5]).all()
  ok = np.ones(m, bool); ok[[3, 5]] = False
  ref = a[ok].astype(np.float64) @ w.astype(np.float64)
  assert np.abs(c[ok] - ref).max() <= 8 * eps * (np.abs(a[ok]).astype(np.float64) @ np.abs(w)).max()


def test_im2col_can_clear_the_absmax_slot_table(lib):
  """EposIm2colArgs.amax_clear (round 4): the plan's slot table is zeroed by the im2col launch
  that opens it -- same columns as without, every word of the table zero afterwards, a count
  beyond the launch's threads is refused."""
  from epos_amd import _lib
  rng = np.random.RandomState(0)
  x = rng.uniform(0, 255, (1, 20, 24, 3)).astype(np.float32)
  X = torch.from_numpy(x).cuda()
  ho, wo, ld = 10, 12, 28
  cols = []
  for clear in (0, 1):
    col = torch.zeros(ho * wo, ld, device='cuda')
    table = torch.full((37 * 64,), 0x3f800000, dtype=torch.int32, device='cuda')
    a = _lib.Im2colArgs(X=_p(X), ldx=3, col=_p(col), ldcol=ld, B=1, Hi=20, Wi=24, Ho=ho, Wo=wo,
                        C=3, stride=2, rate=1, pad=1, preprocess=1,
                        amax_clear=_p(table) if clear else None, amax_words=37 * 64)
    _check(lib.epos_im2col3x3_f32(ctypes.byref(a), None))
    torch.cuda.synchronize()
    cols.append(col)
    assert int(table.abs().sum()) == (0 if clear else 37 * 64 * 0x3f800000)
  assert torch.equal(cols[0], cols[1])
  a.amax_words = ho * wo * ld + 1
  with pytest.raises(_lib.EposError):
    _check(lib.epos_im2col3x3_f32(ctypes.byref(a), None))
