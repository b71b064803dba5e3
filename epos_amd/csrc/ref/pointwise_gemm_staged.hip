// TEST-ONLY translation unit (libepos_hip_ref.so, `python -m epos_amd.build --ref`): the
// register-staged fp32-MFMA GEMM kernels, the first GEMM of this project (round 1). The
// product library routes every GEMM to the fp16-pair kernel or its bf16 x 6 fallback; these
// kernels remain as the fp32-MFMA REFERENCE the accuracy tests compare against
// (tests/test_gpu_layers.py, tests/test_gpu_h2.py, the full-size C2 accuracy test with
// EPOS_GEMM_SPLIT=0) and register themselves with the dispatcher when linked in.
//
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
// Tiling (wave64): BM x 128 block tile (BM = 128 or 64), 4 waves as 2x2, each
// wave a (BM/2) x 64 tile = TM x 2 MFMA tiles of 32x32. K is consumed in steps of
// 32 through double-buffered LDS:
//   A tile  [BM][32(+4 pad)] row-major, filled with coalesced float4 loads along
//           the channel axis (NHWC => K is contiguous); the 36-float row stride
//           makes the per-lane ds_read_b128 of 4 consecutive k conflict-free.
//   W tile  [8][128][4]: the weights are PRE-PACKED on the host into
//           [K/4][Npad][4] so that both the global load and the ds_read_b128 of a
//           lane's 4 consecutive k for one output channel are contiguous.
// MFMA step j of k-group g uses k = 8g + j on lanes 0-31 and k = 8g + 4 + j on
// lanes 32-63 (any pairing of k is valid as long as A and W agree), so one
// ds_read_b128 per operand feeds four MFMAs.
//
// Schedule: an fp32 MFMA occupies the matrix pipe for 64 cycles, so the only job
// of the loop is to never let the pipe drain. The loop is software-rotated: the
// fragments of k-group g+1 are read from LDS before the 16 MFMAs of group g are
// issued, and at the last group of a K tile the register-staged next tile is
// written to the other LDS buffer, the (single) barrier of the tile is crossed
// and the first fragments of the next tile are read -- all in the shadow of the
// last group's MFMAs. Global loads of tile t+1 are issued at the top of tile t.
//
// A launch is GROUPED: up to 8 independent problems (e.g. the four ASPP branches,
// or the three logit heads) share one grid so that small problems still fill the
// 256 CUs.
#include "../pointwise_gemm.h"

namespace epos {
namespace {
template <int BM, bool RELU_IN, bool HAS_RES>
__global__ __launch_bounds__(THREADS) void pointwise_gemm_f32(GroupedArgs ga_) {
  constexpr int TM = BM / 64;                 // MFMA tiles per wave along M
  constexpr int LDS_A_TILE = BM * LDS_A_ROW;
  constexpr int A_LOADS = BM / 32;            // float4 loads per thread per K tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                           // [2][LDS_A_TILE]
  float* Bs = smem + 2 * LDS_A_TILE;          // [2][LDS_B_TILE]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  // ---- which problem / tile -------------------------------------------------
  // The grouped arguments are indexed dynamically (by a block-uniform problem
  // id), so they are read straight from the kernarg segment with scalar loads
  // instead of through a by-value copy (which would be spilled to scratch).
  (void)ga_;
  const GroupedArgs* __restrict__ gp =
      (const GroupedArgs*)__builtin_amdgcn_kernarg_segment_ptr();  // addrspace cast
  // XCD-aware remap (blocks are dispatched round-robin over the 8 XCDs, each with
  // a private 4 MiB L2): XCD x works on one contiguous chunk of the logical tile
  // order (all N tiles of a run of M tiles), so an A tile is fetched into that
  // L2 once and reused by its N tiles, and the packed weights stay L2-resident.
  int bid;
  {
    const int total = gp->tile_start[MAX_GROUP];
    const int raw = blockIdx.x, x = raw & 7, idx = raw >> 3;
    const int q = total >> 3, r = total & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_GROUP; ++i)
    if (i < gp->count && bid >= gp->tile_start[i]) pi = i;
  bid -= gp->tile_start[pi];
  const EposPointwiseArgs p = gp->p[pi];
  const int tiles_n = gp->tiles_n[pi];
  const int npad = gp->npad[pi];
  const int tile_n = bid % tiles_n;
  const int tile_m = bid / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int M = p.M, N = p.N, K = p.K;

  // ---- global -> register staging assignments -------------------------------
  const int c4 = t & 7;                    // float4 column within the A tile row
  const float* arow[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    int m = m0 + (t >> 3) + 32 * i;
    m = m < M ? m : M - 1;                 // clamp (stores are predicated)
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

  // Register staging of the next K tile: named scalars (not arrays) so that they
  // are guaranteed to live in VGPRs.
  float4 ga0, ga1, ga2, ga3, gb0, gb1, gb2, gb3;
  ga0 = ga1 = ga2 = ga3 = make_float4(0.f, 0.f, 0.f, 0.f);
  // Only the LAST K tile can be partial (K % 32 != 0). Its loads come from a
  // clamped, valid address and are zero-filled when written to LDS; every other
  // tile is loaded with no select or branch anywhere near the loads (either would
  // make hipcc wait for the data on the spot instead of a K tile later). The
  // pre-activation ReLU is likewise applied at the LDS write.
  bool g_kin = true;
  auto gload = [&](int kt, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    int ko = kt * BK;
    if (TAIL) {
      g_kin = kt * BK + c4 * 4 < K;
      ko = g_kin ? ko : 0;
    }
    ga0 = *reinterpret_cast<const float4*>(arow[0] + ko);
    ga1 = *reinterpret_cast<const float4*>(arow[1] + ko);
    if constexpr (A_LOADS > 2) {
      ga2 = *reinterpret_cast<const float4*>(arow[2] + ko);
      ga3 = *reinterpret_cast<const float4*>(arow[3] + ko);
    }
    const float* wp = wbase + kt * wstep_tile;
    gb0 = *reinterpret_cast<const float4*>(wp);
    gb1 = *reinterpret_cast<const float4*>(wp + wstep_q2);
    gb2 = *reinterpret_cast<const float4*>(wp + 2 * wstep_q2);
    gb3 = *reinterpret_cast<const float4*>(wp + 3 * wstep_q2);
  };
  auto swrite = [&](int buf, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    auto fix = [&](float4 v) {
      if (TAIL && !g_kin) v = make_float4(0.f, 0.f, 0.f, 0.f);
      return RELU_IN ? relu4(v) : v;
    };
    float* a = As + buf * LDS_A_TILE + (t >> 3) * LDS_A_ROW + c4 * 4;
    *reinterpret_cast<float4*>(a) = fix(ga0);
    *reinterpret_cast<float4*>(a + 32 * LDS_A_ROW) = fix(ga1);
    if constexpr (A_LOADS > 2) {
      *reinterpret_cast<float4*>(a + 64 * LDS_A_ROW) = fix(ga2);
      *reinterpret_cast<float4*>(a + 96 * LDS_A_ROW) = fix(ga3);
    }
    float* b = Bs + buf * LDS_B_TILE + (bq * BN + bn) * 4;
    *reinterpret_cast<float4*>(b) = gb0;
    *reinterpret_cast<float4*>(b + 2 * BN * 4) = gb1;
    *reinterpret_cast<float4*>(b + 4 * BN * 4) = gb2;
    *reinterpret_cast<float4*>(b + 6 * BN * 4) = gb3;
  };

  const int a_frag_off = (wm * (BM / 2) + l31) * LDS_A_ROW + h * 4;
  const int b_frag_off = (h * BN + wn * 64 + l31) * 4;
  float4 fa[TM], fb[2];
  auto read_frags = [&](int buf, int g) {
    const float* a_s = As + buf * LDS_A_TILE + a_frag_off + g * 8;
    const float* b_s = Bs + buf * LDS_B_TILE + b_frag_off + g * 2 * BN * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      fa[i] = *reinterpret_cast<const float4*>(a_s + i * 32 * LDS_A_ROW);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      fb[j] = *reinterpret_cast<const float4*>(b_s + j * 32 * 4);
  };

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (K + BK - 1) / BK;
  gload(0, std::true_type{});
  swrite(0, std::true_type{});
  __syncthreads();
  read_frags(0, 0);

  // One K tile. HAS_NEXT is a compile-time tag so that the steady-state body has
  // no data-dependent control flow around the register staging (which would push
  // the staged tile into scratch memory and serialise the prefetch).
  auto tile = [&](int kt, auto has_next_tag, auto next_tail_tag) {
    constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
    const int buf = kt & 1;
    if (HAS_NEXT) {
      gload(kt + 1, next_tail_tag);
      // keep the prefetch at the top of the tile (hipcc otherwise sinks the loads
      // to their first use, right in front of the barrier)
      __builtin_amdgcn_sched_barrier(0);
    }
    const int kleft = K - kt * BK;         // valid k in this tile (may be < 32)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 ca[TM], cb[2];
#pragma unroll
      for (int i = 0; i < TM; ++i) ca[i] = fa[i];
#pragma unroll
      for (int j = 0; j < 2; ++j) cb[j] = fb[j];
      if (g < 3) {
        read_frags(buf, g + 1);
      } else if (HAS_NEXT) {
        swrite(buf ^ 1, next_tail_tag);
        __syncthreads();
        read_frags(buf ^ 1, 0);
      }
      // Pin the fragment reads of the NEXT k-group in front of this group's MFMAs
      // (hipcc otherwise sinks them behind the MFMAs and then waits for the LDS
      // right at the next group boundary, draining the matrix pipe four times per
      // K tile).
      __builtin_amdgcn_sched_barrier(0);
      if (HAS_NEXT || g * 8 < kleft) {     // wave-uniform: skip all-zero k-groups
        const float* afp = reinterpret_cast<const float*>(ca);
        const float* bfp = reinterpret_cast<const float*>(cb);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                  afp[i * 4 + s], bfp[j * 4 + s], acc[i][j], 0, 0, 0);
      }
    }
  };
  for (int kt = 0; kt + 2 < nk; ++kt)
    tile(kt, std::true_type{}, std::false_type{});
  if (nk >= 2) tile(nk - 2, std::true_type{}, std::true_type{});
  tile(nk - 1, std::false_type{}, std::false_type{});

  // ---- epilogue -------------------------------------------------------------
  if (vec_epilogue_ok(p, HAS_RES)) {
    __syncthreads();                       // every wave is done with the K tiles
    float* ws = smem + wave * (BM / 2) * EP_ROW;
    vec_epilogue<TM, 2, HAS_RES>(ws, &acc[0][0], p, m0 + wm * (BM / 2),
                                 n0 + wn * 64, lane);
    return;
  }
  // scalar path: bias (+ residual) (+ ReLU), predicated 4-byte stores
  // Residual values are fetched with unconditional (clamped) loads, a whole
  // 32x32 tile at a time, so that they are in flight together.
  const bool relu = p.relu != 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
    const int nc = n < N ? n : N - 1;
    const float bias = p.bias ? p.bias[nc] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * h;
      float rv[16];
      if (HAS_RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = mb + (r & 3) + 8 * (r >> 2);
          m = m < M ? m : M - 1;
          rv[r] = p.R[static_cast<int64_t>(m) * p.ldr + nc];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        float v = acc[i][j][r] + bias;
        if (HAS_RES) v += rv[r];
        if (relu) v = fmaxf(v, 0.f);
        if (m < M && n < N) p.C[static_cast<int64_t>(m) * p.ldc + n] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Barrier-free variant ("wave-private"): every wave owns a 32 x 64 output tile
// (1 x 2 MFMA tiles), stages ITS OWN 32 rows of A through a private LDS region
// (the LDS round trip is only the row-major -> k-fragment transposition) and
// reads its W fragments straight from global memory -- the pre-packed
// [K/4][Npad][4] weight layout makes that a contiguous 512 B per half-wave. No
// s_barrier anywhere: waves never wait for each other, a wave whose rows or
// columns are padding simply does less, and the four waves of a workgroup (a
// 128 x 64 tile, stacked along M) only share the L1 hits on W.
// ---------------------------------------------------------------------------
constexpr int WP_ROWS = 32;
constexpr int WP_LDS_TILE = WP_ROWS * LDS_A_ROW;   // floats per buffer per wave

template <bool RELU_IN, bool HAS_RES>
__global__ __launch_bounds__(THREADS) void pointwise_gemm_wp_f32(GroupedArgs ga_) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  float* As = smem + wave * 2 * WP_LDS_TILE;

  (void)ga_;
  const GroupedArgs* __restrict__ gp =
      (const GroupedArgs*)__builtin_amdgcn_kernarg_segment_ptr();  // addrspace cast
  int bid;
  {
    const int total = gp->tile_start[MAX_GROUP];
    const int raw = blockIdx.x, x = raw & 7, idx = raw >> 3;
    const int q = total >> 3, r = total & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_GROUP; ++i)
    if (i < gp->count && bid >= gp->tile_start[i]) pi = i;
  bid -= gp->tile_start[pi];
  const EposPointwiseArgs p = gp->p[pi];
  const int tiles_n = gp->tiles_n[pi];         // 64-column tiles
  const int npad = gp->npad[pi];
  const int tile_n = bid % tiles_n;
  const int tile_m = bid / tiles_n;
  const int M = p.M, N = p.N, K = p.K;
  const int m0 = tile_m * 128 + wave * WP_ROWS, n0 = tile_n * 64;
  if (m0 >= M) return;                          // no barriers: a wave may leave

  const int c4 = lane & 7;
  const float* arow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + (lane >> 3) + 8 * i;
    m = m < M ? m : M - 1;
    int64_t row = m;
    if (p.sub > 1) {
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, rem = m - b * hw;
      const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
      row = (static_cast<int64_t>(b) * p.Hi + yo * p.sub) * p.Wi + xo * p.sub;
    }
    arow[i] = p.A + row * p.lda + c4 * 4;
  }
  const float* wbase = p.Wp + (static_cast<int64_t>(h) * npad + n0 + l31) * 4;
  const int64_t wstep_g = static_cast<int64_t>(2) * npad * 4;    // +1 k-group of 8
  const int64_t wstep_tile = static_cast<int64_t>(8) * npad * 4; // +1 K tile
  const int a_frag_off = l31 * LDS_A_ROW + h * 4;
  const int a_stage_off = (lane >> 3) * LDS_A_ROW + c4 * 4;
  const int nk = (K + BK - 1) / BK;

  auto body = [&](auto tn_tag) {
    constexpr int TN = decltype(tn_tag)::value;    // valid 32-column subtiles
    float4 a0, a1, a2, a3;                         // staged A rows of the next tile
    float4 bc[4][TN], bn[4][TN];                   // W fragments: this / next tile
    bool kin = true;
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    auto gload = [&](int kt, auto tail_tag) {
      constexpr bool TAIL = decltype(tail_tag)::value;
      int ko = kt * BK;
      if (TAIL) {
        kin = kt * BK + c4 * 4 < K;
        ko = kin ? ko : 0;
      }
      a0 = *reinterpret_cast<const float4*>(arow[0] + ko);
      a1 = *reinterpret_cast<const float4*>(arow[1] + ko);
      a2 = *reinterpret_cast<const float4*>(arow[2] + ko);
      a3 = *reinterpret_cast<const float4*>(arow[3] + ko);
      const float* wp = wbase + kt * wstep_tile;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bn[g][j] = *reinterpret_cast<const float4*>(wp + g * wstep_g + j * 32 * 4);
    };
    auto swrite = [&](int buf, auto tail_tag) {
      constexpr bool TAIL = decltype(tail_tag)::value;
      auto fix = [&](float4 v) {
        if (TAIL && !kin) v = make_float4(0.f, 0.f, 0.f, 0.f);
        return RELU_IN ? relu4(v) : v;
      };
      float* a = As + buf * WP_LDS_TILE + a_stage_off;
      *reinterpret_cast<float4*>(a) = fix(a0);
      *reinterpret_cast<float4*>(a + 8 * LDS_A_ROW) = fix(a1);
      *reinterpret_cast<float4*>(a + 16 * LDS_A_ROW) = fix(a2);
      *reinterpret_cast<float4*>(a + 24 * LDS_A_ROW) = fix(a3);
    };
    float4 fa;
    auto read_frag = [&](int buf, int g) {
      fa = *reinterpret_cast<const float4*>(As + buf * WP_LDS_TILE + a_frag_off + g * 8);
    };
    auto mfma_steps = [&](const float4& ca, int g, int s0, int s1) {
      const float* afp = reinterpret_cast<const float*>(&ca);
#pragma unroll
      for (int sidx = s0; sidx < s1; ++sidx)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
              afp[sidx], reinterpret_cast<const float*>(&bc[g][j])[sidx], acc[j],
              0, 0, 0);
    };

    gload(0, std::true_type{});
    swrite(0, std::true_type{});
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < TN; ++j) bc[g][j] = bn[g][j];
    read_frag(0, 0);

    auto tile = [&](int kt, auto has_next_tag, auto next_tail_tag) {
      constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
      const int buf = kt & 1;
      if (HAS_NEXT) {
        gload(kt + 1, next_tail_tag);
        __builtin_amdgcn_sched_barrier(0);       // keep the prefetch up here
      }
      const int kleft = K - kt * BK;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 ca = fa;
        const bool live = HAS_NEXT || g * 8 < kleft;
        if (g < 3) {
          read_frag(buf, g + 1);
          __builtin_amdgcn_sched_barrier(0);     // reads first, then the MFMAs
          if (live) mfma_steps(ca, g, 0, 4);
        } else {
          if (HAS_NEXT) swrite(buf ^ 1, next_tail_tag);
          __builtin_amdgcn_sched_barrier(0);
          if (live) mfma_steps(ca, g, 0, 2);
          if (HAS_NEXT) read_frag(buf ^ 1, 0);   // own LDS region: no barrier
          __builtin_amdgcn_sched_barrier(0);
          if (live) mfma_steps(ca, g, 2, 4);
        }
      }
      if (HAS_NEXT) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < TN; ++j) bc[g][j] = bn[g][j];
      }
    };
    for (int kt = 0; kt + 2 < nk; ++kt)
      tile(kt, std::true_type{}, std::false_type{});
    if (nk >= 2) tile(nk - 2, std::true_type{}, std::true_type{});
    tile(nk - 1, std::false_type{}, std::false_type{});

    // ---- epilogue ---------------------------------------------------------
    if (vec_epilogue_ok(p, HAS_RES)) {       // private LDS region: no barrier
      vec_epilogue<1, TN, HAS_RES>(As, acc, p, m0, n0, lane);
      return;
    }
    const bool relu = p.relu != 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + j * 32 + l31;
      const int nc = n < N ? n : N - 1;
      const float bias = p.bias ? p.bias[nc] : 0.f;
      const int mb = m0 + 4 * h;
      float rv[16];
      if (HAS_RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = mb + (r & 3) + 8 * (r >> 2);
          m = m < M ? m : M - 1;
          rv[r] = p.R[static_cast<int64_t>(m) * p.ldr + nc];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        float v = acc[j][r] + bias;
        if (HAS_RES) v += rv[r];
        if (relu) v = fmaxf(v, 0.f);
        if (m < M && n < N) p.C[static_cast<int64_t>(m) * p.ldc + n] = v;
      }
    }
  };
  if (n0 + 32 < N) {
    body(std::integral_constant<int, 2>{});
  } else {
    body(std::integral_constant<int, 1>{});      // second 32 columns are padding
  }
}
template <int BM, bool RELU_IN, bool HAS_RES>
int launch_grouped_t(const GroupedArgs& g, int total, hipStream_t s) {
  constexpr int LDS_A_TILE = BM * LDS_A_ROW;
  constexpr size_t LDS_MAX = 160 * 1024;
  size_t lds = sizeof(float) * (2 * LDS_A_TILE + 2 * LDS_B_TILE);
  static LdsAttrOnce once;
  {
    const int rc = ensure_dynamic_lds(
        once, reinterpret_cast<const void*>(pointwise_gemm_f32<BM, RELU_IN, HAS_RES>),
        static_cast<int>(LDS_MAX), "hipFuncSetAttribute(pointwise_gemm_f32)");
    if (rc) return rc;
  }
  // Workgroup placement: the dispatcher packs workgroups onto a CU while its
  // resources last, so a grid that fits "3 per CU" leaves CUs idle. Request just
  // enough extra LDS that at most ceil(grid / 256) workgroups fit on one CU; the
  // grid is then spread over all 256 CUs.
  static const int spread = [] {
    const char* e = getenv("EPOS_GEMM_SPREAD");
    return e ? atoi(e) : 1;
  }();
  if (spread) {
    int per_cu = (total + 255) / 256;
    if (spread > 1 && per_cu > spread) per_cu = spread;   // EPOS_GEMM_SPREAD=k: cap
    const size_t cap = (LDS_MAX / per_cu) & ~static_cast<size_t>(1023);
    if (cap > lds) lds = cap;
  }
  static const int lds_kb = [] {            // EPOS_GEMM_LDS_KB=k: fixed LDS request
    const char* e = getenv("EPOS_GEMM_LDS_KB");
    return e ? atoi(e) : 0;
  }();
  if (lds_kb > 0 && static_cast<size_t>(lds_kb) * 1024 > lds && lds_kb <= 160)
    lds = static_cast<size_t>(lds_kb) * 1024;
  hipLaunchKernelGGL((pointwise_gemm_f32<BM, RELU_IN, HAS_RES>), dim3(total),
                     dim3(THREADS), lds, s, g);
  return launch_status("pointwise_gemm_f32");
}

template <int BM>
int launch_grouped(const EposPointwiseArgs* args, int count, hipStream_t s) {
  GroupedArgs g;
  g.count = count;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    g.p[i] = args[i];
    g.npad[i] = static_cast<int>(round_up(args[i].N, BN));
    g.tiles_n[i] = g.npad[i] / BN;
    g.tile_start[i] = total;
    total += static_cast<int>(ceil_div(args[i].M, BM)) * g.tiles_n[i];
  }
  for (int i = count; i <= MAX_GROUP; ++i) g.tile_start[i] = total;
  const bool relu_in = args[0].relu_in != 0, has_res = args[0].R != nullptr;
  if (relu_in) {
    return has_res ? launch_grouped_t<BM, true, true>(g, total, s)
                   : launch_grouped_t<BM, true, false>(g, total, s);
  }
  return has_res ? launch_grouped_t<BM, false, true>(g, total, s)
                 : launch_grouped_t<BM, false, false>(g, total, s);
}

template <bool RELU_IN, bool HAS_RES>
int launch_wp_t(const GroupedArgs& g, int total, hipStream_t s) {
  const size_t lds = sizeof(float) * 4 * 2 * WP_LDS_TILE;
  hipLaunchKernelGGL((pointwise_gemm_wp_f32<RELU_IN, HAS_RES>), dim3(total),
                     dim3(THREADS), lds, s, g);
  return launch_status("pointwise_gemm_wp_f32");
}

int launch_grouped_wp(const EposPointwiseArgs* args, int count, hipStream_t s) {
  GroupedArgs g;
  g.count = count;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    g.p[i] = args[i];
    g.npad[i] = static_cast<int>(round_up(args[i].N, BN));
    g.tiles_n[i] = static_cast<int>(ceil_div(args[i].N, 64));
    g.tile_start[i] = total;
    total += static_cast<int>(ceil_div(args[i].M, 128)) * g.tiles_n[i];
  }
  for (int i = count; i <= MAX_GROUP; ++i) g.tile_start[i] = total;
  const bool relu_in = args[0].relu_in != 0, has_res = args[0].R != nullptr;
  if (relu_in)
    return has_res ? launch_wp_t<true, true>(g, total, s)
                   : launch_wp_t<true, false>(g, total, s);
  return has_res ? launch_wp_t<false, true>(g, total, s)
                 : launch_wp_t<false, false>(g, total, s);
}

}  // namespace

// Register-staged kernels (pre-activation ReLU on the way into LDS):
// EPOS_GEMM_TILE_M=64|128 and EPOS_GEMM_WP=0|1 override the choices for tuning.
// 128-row tiles only when they alone give every CU >= 2 workgroups; the
// barrier-free kernel tiles N in steps of 64, so it takes the groups whose
// problems all have N <= 64.
int launch_grouped_staged(const EposPointwiseArgs* args, int count, hipStream_t s) {
  for (int i = 0; i < count; ++i)
    if (args[i].c_amax) {
      set_error("launch_grouped_staged: c_amax is not available with the register-staged kernels");
      return EPOS_E_INVALID;
    }
  static const int forced = [] {
    const char* e = getenv("EPOS_GEMM_TILE_M");
    return e ? atoi(e) : 0;
  }();
  static const int use_wp = [] {
    const char* e = getenv("EPOS_GEMM_WP");
    return e ? atoi(e) : -1;
  }();
  int max_n = 0;
  int64_t tiles128 = 0;
  for (int i = 0; i < count; ++i) {
    max_n = args[i].N > max_n ? args[i].N : max_n;
    tiles128 += ceil_div(args[i].M, 128) * ceil_div(args[i].N, BN);
  }
  if (use_wp == 1 || (use_wp < 0 && max_n <= 64)) return launch_grouped_wp(args, count, s);
  const bool big = forced ? forced == 128 : tiles128 >= 512;
  if (big) return launch_grouped<128>(args, count, s);
  return launch_grouped<64>(args, count, s);
}

namespace {
const int registered_staged = (fp32_mfma_ref().staged = &launch_grouped_staged, 0);
}  // namespace
}  // namespace epos
