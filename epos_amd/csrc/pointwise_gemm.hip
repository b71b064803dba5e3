// 1x1 convolution (pointwise contraction) for gfx950: the C entry points, argument
// validation, host-side weight packing of the plain layout, the M <= 8 GEMV, and the routing
//
//   C[m, n] = act( sum_k A[row(m), k] * W[k, n] + bias[n] (+ R[m, n]) )
//
// Replaces slim.conv2d(kernel=1) + folded BatchNorm (+ residual) (+ ReLU):
// net_xception.py:167-182,296-302, model.py:223-224,237,257-258,349-352,449-456.
// 98 % of the network's FLOPs go through the GEMM behind it:
//   * pointwise_gemm_h2.hip     fp16-pair kernel (three fp16 MFMA products per fp32 product),
//                               the default whenever the caller supplies Wh + a bound for A;
//   * pointwise_gemm_split.hip  bf16 x 6 kernel, the fallback (Ws);
//   * ref/*.hip                 the fp32-MFMA kernels of rounds 1-2 (v_mfma_f32_32x32x2_f32:
//                               exact fp32 multiply-add chains) -- NOT in the product library
//                               since round 6: they are linked into libepos_hip_ref.so only,
//                               where the accuracy tests compare the two kernels above
//                               against them, and register themselves through
//                               fp32_mfma_ref() below. In the product library a problem that
//                               brings neither Wh nor Ws is refused loudly.
// A launch is GROUPED: up to 8 independent problems (e.g. the four ASPP branches,
// or the three logit heads) share one grid so that small problems still fill the
// 256 CUs.
#include "pointwise_gemm.h"

namespace epos {

Fp32MfmaRef& fp32_mfma_ref() {
  static Fp32MfmaRef hooks = {nullptr, nullptr};
  return hooks;
}

namespace {

int no_fp32_mfma(const char* what) {
  set_error("%s: this problem brings neither fp16-pair (Wh + a bound for A) nor bf16 x 6 (Ws) "
            "weights, and this build of libepos_hip.so carries no fp32-MFMA kernel (they are "
            "test-only since round 6: libepos_hip_ref.so, `python -m epos_amd.build --ref`, "
            "EPOS_HIP_LIB=<that file>)", what);
  return EPOS_E_INVALID;
}

// ---------------------------------------------------------------------------
// M <= 8 rows (the image-pooling branch, model.py:223-224, M = batch): a GEMV.
// Block = 64 output channels x 16 K-slices (1024 threads); every thread keeps eight
// independent loads in flight (the first version, 4 slices and one dependent load
// per iteration, took 62 us for this 1 MFLOP); fixed-order LDS reduction.
// ---------------------------------------------------------------------------
constexpr int GEMV_SLICES = 16;

__global__ __launch_bounds__(64 * GEMV_SLICES) void pointwise_gemv_f32(EposPointwiseArgs p,
                                                                       int npad) {
  __shared__ float part[GEMV_SLICES][64];
  const int t = threadIdx.x;
  const int n = blockIdx.x * 64 + (t & 63);
  const int ks = t >> 6;
  const int m = blockIdx.y;
  const float* a = p.A + static_cast<int64_t>(m) * p.lda;
  const int kq = p.K / 4;
  float acc = 0.f;
  int q = ks;
  for (; q + 7 * GEMV_SLICES < kq; q += 8 * GEMV_SLICES) {
    float4 av[8], wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      av[u] = *reinterpret_cast<const float4*>(a + (q + u * GEMV_SLICES) * 4);
      wv[u] = *reinterpret_cast<const float4*>(
          p.Wp + (static_cast<int64_t>(q + u * GEMV_SLICES) * npad + n) * 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (p.relu_in) av[u] = relu4(av[u]);
      acc = fmaf(av[u].x, wv[u].x, acc);
      acc = fmaf(av[u].y, wv[u].y, acc);
      acc = fmaf(av[u].z, wv[u].z, acc);
      acc = fmaf(av[u].w, wv[u].w, acc);
    }
  }
  for (; q < kq; q += GEMV_SLICES) {
    float4 av = *reinterpret_cast<const float4*>(a + q * 4);
    if (p.relu_in) av = relu4(av);
    const float4 wv = *reinterpret_cast<const float4*>(
        p.Wp + (static_cast<int64_t>(q) * npad + n) * 4);
    acc = fmaf(av.x, wv.x, acc);
    acc = fmaf(av.y, wv.y, acc);
    acc = fmaf(av.z, wv.z, acc);
    acc = fmaf(av.w, wv.w, acc);
  }
  part[ks][t & 63] = acc;
  __syncthreads();
  if (ks == 0 && n < p.N) {
    float v = part[0][t];
#pragma unroll
    for (int i = 1; i < GEMV_SLICES; ++i) v += part[i][t];
    if (p.bias) v += p.bias[n];
    if (p.R) v += p.R[static_cast<int64_t>(m) * p.ldr + n];
    if (p.relu) v = fmaxf(v, 0.f);
    p.C[static_cast<int64_t>(m) * p.ldc + n] = v;
  }
}

int validate(const EposPointwiseArgs* a) {
  EPOS_REQUIRE(a->A && a->Wp && a->C, "null pointer");
  EPOS_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "empty problem");
  EPOS_REQUIRE(a->K % 4 == 0 && a->lda % 4 == 0, "K and lda must be multiples of 4");
  EPOS_REQUIRE((reinterpret_cast<uintptr_t>(a->A) & 15) == 0, "A must be 16-byte aligned");
  EPOS_REQUIRE(a->sub >= 1, "sub must be >= 1");
  // a pre-split A operand (fp16 pairs written by the depthwise kernel) can only be read by
  // the fp16-pair kernel, with the scale its producer used
  EPOS_REQUIRE(!a->a_presplit || (a->Wh && a->a_amax && a->relu_in == 0 && a->M > 8 &&
                                  a->sub == 1),
               "a_presplit needs Wh, a_amax, relu_in == 0, sub == 1 and M > 8");
  if (a->c_amax) {
    // the absmax of the output is taken in the float4 epilogue of the LDS-DMA kernels
    bool ok = a->relu_in == 0 && a->M > 8 && (a->N & 3) == 0 && (a->ldc & 3) == 0 &&
              (reinterpret_cast<uintptr_t>(a->C) & 15) == 0;
    if (a->R) ok = ok && (a->ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(a->R) & 15) == 0;
    EPOS_REQUIRE(ok, "c_amax needs N, ldc (ldr) multiples of 4, 16-byte aligned C (R), "
                     "relu_in == 0 and M > 8");
  }
  return EPOS_OK;
}

}  // namespace
}  // namespace epos

extern "C" int64_t epos_pack_pointwise_weights(const float* w_kn, int K, int N,
                                               float* dst) {
  using namespace epos;
  const int64_t kg = ceil_div(K, BK) * (BK / 4);
  const int64_t npad = round_up(N, BN);
  const int64_t total = kg * npad * 4;
  if (!dst) return total;
  for (int64_t q = 0; q < kg; ++q)
    for (int64_t n = 0; n < npad; ++n)
      for (int e = 0; e < 4; ++e) {
        const int64_t k = q * 4 + e;
        dst[(q * npad + n) * 4 + e] =
            (k < K && n < N) ? w_kn[k * static_cast<int64_t>(N) + n] : 0.f;
      }
  return total;
}

static int grouped_impl(const EposPointwiseArgs* args, int count, void* stream);

extern "C" int epos_pointwise_conv_grouped_f32(const EposPointwiseArgs* args,
                                               int count, void* stream) {
  using namespace epos;
  EPOS_REQUIRE(args && count >= 1 && count <= MAX_GROUP, "1..8 problems per group");
  for (int i = 0; i < count; ++i) {
    const EposPointwiseArgs& a = args[i];
    EPOS_REQUIRE(a.reserved0 == 0, "EposPointwiseArgs.reserved0 must be 0 (ABI 7)");
    if (a.col_sums)
      EPOS_REQUIRE(!a.R && a.N % 4 == 0 && a.col_ld % 4 == 0 && a.col_ld >= a.N &&
                       (reinterpret_cast<uintptr_t>(a.col_sums) & 15) == 0 &&
                       (a.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 &&
                       h2_eligible(args, count),
                   "col_sums needs the fp16-pair kernel (Wh, a bound for A), no residual, "
                   "N, ldc, col_ld multiples of 4 and 16-byte aligned C / col_sums");
  }
  return grouped_impl(args, count, stream);
}

static int grouped_impl(const EposPointwiseArgs* args, int count, void* stream) {
  using namespace epos;
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int i = 0; i < count; ++i) {
    const int rc = validate(&args[i]);
    if (rc) return rc;
    EPOS_REQUIRE((args[i].relu_in != 0) == (args[0].relu_in != 0) &&
                 (args[i].R != nullptr) == (args[0].R != nullptr),
                 "problems of one group must agree on relu_in / residual");
  }
  if (count == 1 && args[0].M <= 8 && args[0].sub == 1) {
    const int npad = static_cast<int>(round_up(args[0].N, BN));
    hipLaunchKernelGGL(pointwise_gemv_f32, dim3(npad / 64, args[0].M),
                       dim3(64 * GEMV_SLICES), 0, s, args[0], npad);
    return launch_status("pointwise_gemv_f32");
  }
  // Split-operand kernel on the bf16 matrix pipe (pointwise_gemm_split.hip): fp32 in,
  // fp32 out, error not above the fp32 MFMA's; taken whenever the caller supplied the
  // split-packed weights. EPOS_GEMM_SPLIT=0 keeps everything on the fp32 MFMA kernels.
  // fp16-pair kernel (pointwise_gemm_h2.hip, round 3): the same contract with half the
  // matrix-pipe work; taken whenever the caller supplied the fp16-pair weights.
  // EPOS_GEMM_H2=0 falls through to the bf16 x 6 kernel.
  if (h2_eligible(args, count)) return launch_grouped_h2(args, count, s);
  for (int i = 0; i < count; ++i)
    EPOS_REQUIRE(!args[i].a_presplit,
                 "a pre-split A operand needs the fp16-pair kernel (every problem of the "
                 "group with Wh; EPOS_GEMM_H2 / EPOS_GEMM_SPLIT not 0)");
  if (split_eligible(args, count)) return launch_grouped_split(args, count, s);
  // Neither fp16-pair nor bf16 x 6 weights: the fp32-MFMA kernels, if this is the test build
  // (ref/pointwise_gemm_dma.hip: LDS-DMA kernel for every problem without a pre-activation
  // ReLU, EPOS_GEMM_DMA=0 disables it; ref/pointwise_gemm_staged.hip: register-staged
  // kernels otherwise).
  const Fp32MfmaRef& ref = fp32_mfma_ref();
  static const int use_dma = [] {
    const char* e = getenv("EPOS_GEMM_DMA");
    return e ? atoi(e) : -1;
  }();
  if (args[0].relu_in == 0 && use_dma != 0 && ref.dma)
    return ref.dma(args, count, s, nullptr, nullptr);
  if (ref.staged) return ref.staged(args, count, s);
  return no_fp32_mfma("epos_pointwise_conv_grouped_f32");
}

extern "C" int epos_conv3x3_f32(const EposConv3x3Args* a, void* stream) {
  using namespace epos;
  EPOS_REQUIRE(a && a->X && a->Wp && a->Y, "null pointer");
  EPOS_REQUIRE(a->Cin > 0 && a->Cin % BK == 0, "Cin must be a multiple of 32");
  EPOS_REQUIRE(a->ldx % 4 == 0 && a->ldx >= a->Cin, "ldx: multiple of 4, >= Cin");
  EPOS_REQUIRE((reinterpret_cast<uintptr_t>(a->X) & 15) == 0, "X must be 16-byte aligned");
  EPOS_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Cout > 0, "empty problem");
  EPOS_REQUIRE((a->stride == 1 || a->stride == 2) && a->rate >= 1, "stride 1|2, rate >= 1");
  const int ho = (a->H - 1) / a->stride + 1, wo = (a->W - 1) / a->stride + 1;
  EPOS_REQUIRE(static_cast<int64_t>(a->B) * a->H * a->W < (1LL << 31), "too many pixels");
  EposPointwiseArgs p = {};
  p.A = a->X; p.lda = a->ldx;
  p.Wp = a->Wp; p.bias = a->bias;
  p.R = nullptr; p.ldr = 0;
  p.C = a->Y; p.ldc = a->ldy;
  p.M = a->B * ho * wo; p.N = a->Cout; p.K = 9 * a->Cin;
  p.relu = a->relu; p.relu_in = 0; p.sub = a->stride;
  p.Ho = ho; p.Wo = wo; p.Hi = a->H; p.Wi = a->W;
  p.Ws = a->Ws;
  p.Wh = a->Wh; p.a_amax = a->x_amax; p.c_amax = a->y_amax;
  {
    const int rc = validate(&p);
    if (rc) return rc;
  }
  const int cin = a->Cin, rate = a->rate;
  if (a->Wh && h2_eligible(&p, 1))
    return launch_grouped_h2(&p, 1, static_cast<hipStream_t>(stream), &cin, &rate);
  // split-operand kernel when the split-packed weights came along (K steps of 16
  // channels inside one tap: Cin % 32 == 0 covers it)
  if (a->Ws && split_eligible(&p, 1))
    return launch_grouped_split(&p, 1, static_cast<hipStream_t>(stream), &cin, &rate);
  if (fp32_mfma_ref().dma)
    return fp32_mfma_ref().dma(&p, 1, static_cast<hipStream_t>(stream), &cin, &rate);
  return no_fp32_mfma("epos_conv3x3_f32");
}

extern "C" int epos_separable_conv_f32(const EposSepConvArgs* a, void* stream) {
  using namespace epos;
  EPOS_REQUIRE(a && a->dw.X && a->dw.w9c && a->dw.bias && a->dw.Y, "null pointer");
  const EposDepthwiseArgs& d = a->dw;
  const EposPointwiseArgs& p = a->pw;
  EPOS_REQUIRE(p.A == d.Y && p.lda == d.ldy && p.K == d.C,
               "the pointwise conv must read the depthwise output");
  EPOS_REQUIRE(static_cast<int64_t>(p.M) == static_cast<int64_t>(d.B) * d.Ho * d.Wo,
               "pw.M must be the number of depthwise output pixels");
  EPOS_REQUIRE((d.y_h2 != 0) == (p.a_presplit != 0),
               "dw.y_h2 and pw.a_presplit describe the same intermediate");
  const int rc = validate(&p);
  if (rc) return rc;
  // the two launches (the single-launch forms of rounds 2 and 4 measured slower and were
  // removed in round 5: include/epos_hip.h)
  const int rd = epos_depthwise3x3_f32(&d, stream);
  if (rd) return rd;
  return epos_pointwise_conv_f32(&p, stream);
}

extern "C" int epos_pointwise_conv_f32(const EposPointwiseArgs* a, void* stream) {
  return epos_pointwise_conv_grouped_f32(a, 1, stream);
}

