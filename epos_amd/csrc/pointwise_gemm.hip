// 1x1 convolution (pointwise contraction) as an fp32 MFMA GEMM for gfx950.
//
//   C[m, n] = act( sum_k A[row(m), k] * W[k, n] + bias[n] (+ R[m, n]) )
//
// Replaces slim.conv2d(kernel=1) + folded BatchNorm (+ residual) (+ ReLU):
// net_xception.py:167-182,296-302, model.py:223-224,237,257-258,349-352,449-456.
// 98 % of the network's FLOPs go through this kernel.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32 multiply-add chains (the
// reference computes in fp32), 64 FLOP/clk/SIMD = the 157.3 TFLOP/s fp32 roof.
//
// Tiling (wave64): 128x128 block tile, 4 waves as 2x2, each wave a 64x64 tile =
// 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs). K is consumed in steps of 32
// through double-buffered LDS:
//   A tile  [128][32(+4 pad)] row-major, filled with coalesced float4 loads along
//           the channel axis (NHWC => K is contiguous); the 36-float row stride
//           makes the per-lane ds_read_b128 of 4 consecutive k conflict-free.
//   W tile  [8][128][4]: the weights are PRE-PACKED on the host into
//           [K/4][Npad][4] so that both the global load and the ds_read_b128 of a
//           lane's 4 consecutive k for one output channel are contiguous.
// MFMA step j of k-group g uses k = 8g + j on lanes 0-31 and k = 8g + 4 + j on
// lanes 32-63 (any pairing of k is valid as long as A and W agree), so one
// ds_read_b128 per operand feeds four MFMAs.
#include "common.h"

namespace epos {
namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 32;
constexpr int LDS_A_ROW = BK + 4;                  // floats
constexpr int LDS_A_TILE = BM * LDS_A_ROW;         // floats per buffer
constexpr int LDS_B_TILE = (BK / 4) * BN * 4;      // floats per buffer
constexpr int THREADS = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f),
                     fmaxf(v.w, 0.f));
}

template <bool RELU_IN>
__global__ __launch_bounds__(THREADS) void pointwise_gemm_f32(EposPointwiseArgs p,
                                                               int tiles_n,
                                                               int npad) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][LDS_A_TILE]
  float* Bs = smem + 2 * LDS_A_TILE;      // [2][LDS_B_TILE]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  const int bid = blockIdx.x;
  const int tile_n = bid % tiles_n;
  const int tile_m = bid / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- global -> register staging assignments -------------------------------
  const int c4 = t & 7;                    // float4 column within the A tile row
  const float* arow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + (t >> 3) + 32 * i;
    m = m < p.M ? m : p.M - 1;             // clamp (stores are predicated)
    int64_t row = m;
    if (p.sub > 1) {
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, rem = m - b * hw;
      const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
      row = (static_cast<int64_t>(b) * p.Hi + yo * p.sub) * p.Wi + xo * p.sub;
    }
    arow[i] = p.A + row * p.lda + c4 * 4;
  }
  const int bn = t & 127;                  // output channel within the W tile
  const int bq = t >> 7;                   // k-group (of 4) 0..1, +2 per i
  const float* wbase = p.Wp + (static_cast<int64_t>(bq) * npad + n0 + bn) * 4;
  const int64_t wstep_q2 = static_cast<int64_t>(2) * npad * 4;   // +2 k-groups
  const int64_t wstep_tile = static_cast<int64_t>(8) * npad * 4; // +1 K tile

  float4 ga[4], gb[4];
  auto gload = [&](int kt) {
    const int k = kt * BK + c4 * 4;
    const bool kin = k < p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kin) v = *reinterpret_cast<const float4*>(arow[i] + kt * BK);
      ga[i] = RELU_IN ? relu4(v) : v;
    }
    const float* wp = wbase + kt * wstep_tile;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      gb[i] = *reinterpret_cast<const float4*>(wp + i * wstep_q2);
  };
  auto swrite = [&](int buf) {
    float* a = As + buf * LDS_A_TILE + (t >> 3) * LDS_A_ROW + c4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(a + 32 * i * LDS_A_ROW) = ga[i];
    float* b = Bs + buf * LDS_B_TILE + (bq * BN + bn) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(b + i * 2 * BN * 4) = gb[i];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  swrite(0);
  __syncthreads();

  const int a_frag_off = (wm * 64 + l31) * LDS_A_ROW + h * 4;
  const int b_frag_off = (h * BN + wn * 64 + l31) * 4;

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const float* a_s = As + buf * LDS_A_TILE + a_frag_off;
    const float* b_s = Bs + buf * LDS_B_TILE + b_frag_off;
    const int kleft = p.K - kt * BK;       // valid k in this tile (may be < 32)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g * 8 < kleft) {                 // wave-uniform: skip all-zero k-groups
        float4 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          af[i] = *reinterpret_cast<const float4*>(a_s + i * 32 * LDS_A_ROW + g * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bf[j] = *reinterpret_cast<const float4*>(b_s + (g * 2 * BN + j * 32) * 4);
        const float* afp = reinterpret_cast<const float*>(af);
        const float* bfp = reinterpret_cast<const float*>(bf);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                  afp[i * 4 + s], bfp[j * 4 + s], acc[i][j], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) swrite(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias (+ residual) (+ ReLU), predicated stores -------------
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
    if (n >= p.N) continue;
    const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M) {
          float v = acc[i][j][r] + bias;
          if (p.R) v += p.R[static_cast<int64_t>(m) * p.ldr + n];
          if (p.relu) v = fmaxf(v, 0.f);
          p.C[static_cast<int64_t>(m) * p.ldc + n] = v;
        }
      }
    }
  }
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

extern "C" int epos_pointwise_conv_f32(const EposPointwiseArgs* a, void* stream) {
  using namespace epos;
  EPOS_REQUIRE(a && a->A && a->Wp && a->C, "null pointer");
  EPOS_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "empty problem");
  EPOS_REQUIRE(a->K % 4 == 0 && a->lda % 4 == 0, "K and lda must be multiples of 4");
  EPOS_REQUIRE((reinterpret_cast<uintptr_t>(a->A) & 15) == 0, "A must be 16-byte aligned");
  EPOS_REQUIRE(a->sub >= 1, "sub must be >= 1");
  const int npad = static_cast<int>(round_up(a->N, BN));
  const int tiles_n = npad / BN;
  const int64_t tiles_m = ceil_div(a->M, BM);
  const size_t lds = sizeof(float) * (2 * LDS_A_TILE + 2 * LDS_B_TILE);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pointwise_gemm_f32<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pointwise_gemm_f32<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid(static_cast<unsigned>(tiles_m * tiles_n));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->relu_in)
    hipLaunchKernelGGL(pointwise_gemm_f32<true>, grid, dim3(THREADS), lds, s, *a,
                       tiles_n, npad);
  else
    hipLaunchKernelGGL(pointwise_gemm_f32<false>, grid, dim3(THREADS), lds, s, *a,
                       tiles_n, npad);
  return launch_status("pointwise_gemm_f32");
}
