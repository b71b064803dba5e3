// LDS-DMA pointwise GEMM kernel for gfx950 on the fp32 matrix pipe (pointwise_gemm_dma_f32).
// Tiling, operand layouts and the C-level contract are those of pointwise_gemm.hip; DESIGN.md
// ("The fp32-MFMA kernel") holds the measurements behind every choice below. (A persistent
// stream-K variant on the same ring lived here through round 4: correct, tested, slower end
// to end -- removed in round 5, history up to commit 1b05025.)
#include "../pointwise_gemm.h"

namespace epos {
namespace {

// ---------------------------------------------------------------------------
// LDS-DMA variant ("dma"): the same 64 x 128 block tile and MFMA schedule as
// pointwise_gemm_f32<64>, but the K tiles travel global -> LDS with
// global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass) into a THREE-stage
// ring, so the loads of tile t+2 are issued at the top of tile t and are only
// waited for (counted vmcnt, never 0 in steady state) a full tile later -- about
// two K-tile times of latency tolerance instead of 0.6. One raw s_barrier per
// tile: it (i) makes tile t+1 (landed: every wave waited for its own pieces)
// visible to all waves and (ii) frees stage t % 3 for the loads of tile t+3.
//
// The LDS-DMA destination is wave-uniform base + lane * 16 B, i.e. the LDS image
// of a piece is lane-linear; the A tile therefore has unpadded 128-byte rows and
// is made conflict-free by an XOR swizzle applied to the per-lane SOURCE address
// (slot c' of row r holds chunk c' ^ ((r >> 1) & 7)) and to the fragment reads.
// The W tile keeps the packed [8][128][4] image (already conflict-free).
// hipcc cannot see asm LDS-DMA, so it neither waits for it before every ds_read
// (which it would do for the builtin) nor counts it: the vmcnt waits below are
// explicit, and no ordinary global load is live while a DMA is in flight.
// The partial last K tile takes its out-of-range chunks from a zero block.
// ---------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) float g_zero_chunk[4] = {0.f, 0.f, 0.f, 0.f};

// Build with -DEPOS_GEMM_TRACE (tools/gemm_trace.py) to record per-workgroup phase
// timestamps (s_memrealtime, 100 MHz) of the LDS-DMA kernel.
#ifdef EPOS_GEMM_TRACE
__device__ unsigned long long g_trace[8192 * 8];
__device__ unsigned long long g_trace_units[512 * 32];   // stream-K: end time of each unit
#define EPOS_TRACE(slot)                                                     \
  do {                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 8192)                               \
      g_trace[blockIdx.x * 8 + (slot)] = wall_clock64();                     \
  } while (0)
#else
#define EPOS_TRACE(slot) do {} while (0)
#endif

constexpr int DMA_B_BYTES = (BK / 4) * BN * 16;          // 16384
constexpr int DMA_A_BYTES = 64 * BK * 4;                 // 8192
constexpr int DMA_STAGE_BYTES = DMA_A_BYTES + DMA_B_BYTES;
constexpr int DMA_STAGES = 3;
constexpr int DMA_LDS_BYTES = DMA_STAGES * DMA_STAGE_BYTES;   // 73728

// LAYOUT 0: 64 x 128 tile, waves 2 x 2.   LAYOUT 1: 128 x 64 tile, waves 4 x 1 (every
// problem of the launch has N <= 64: a 128-wide tile would waste half of its MFMAs).
// The per-wave tile (32 x 64), the six DMA pieces per wave and K tile and the 24 KB
// stage are the same in both.
// CONV: implicit GEMM of a dense 3x3 conv with Cin % 32 == 0 (conv2d_same,
// external/slim/nets/resnet_utils.py:77-122: stride 1 = 'SAME' with dilation `rate`,
// stride 2 = explicit pad `rate` + VALID; conv1_2 net_xception.py:462-463, the root and
// bottleneck 3x3 convs of net_resnet_v1_beta.py): K tile kt is channel block
// kt % (Cin/32) of tap kt / (Cin/32), i.e. the A rows are the input pixels
// (y*stride + dy*rate, x*stride + dx*rate) -- the im2col matrix only ever exists as
// LDS tiles. Taps outside the image come from a zero block (a per-lane source
// select, as for the partial last K tile).
template <bool HAS_RES, int LAYOUT, bool CONV, bool SINGLE>
__global__ __launch_bounds__(THREADS) void pointwise_gemm_dma_f32(GroupedArgs ga_) {
  constexpr int BM_ = LAYOUT == 0 ? 64 : 128;
  constexpr int BN_ = LAYOUT == 0 ? 128 : 64;
  constexpr int NA = BM_ / 32;               // A pieces per wave and K tile
  constexpr int NW = BN_ / 32;               // W pieces per wave and K tile
  constexpr int B_BYTES = (BK / 4) * BN_ * 16;
  static_assert(NA + NW == 6 && B_BYTES + BM_ * BK * 4 == DMA_STAGE_BYTES, "stage");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm = LAYOUT == 0 ? wave >> 1 : wave, wn = LAYOUT == 0 ? wave & 1 : 0;
  const int l31 = lane & 31, h = lane >> 5;
  EPOS_TRACE(0);
#ifdef EPOS_GEMM_TRACE
  if (t == 0 && blockIdx.x < 8192) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_trace[blockIdx.x * 8 + 4] = (static_cast<unsigned long long>(xcc) << 32) | hw;
    g_trace[blockIdx.x * 8 + 5] = clock64();
  }
#endif

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
  // SINGLE (one problem per launch, the common case): every argument is read from a
  // fixed kernarg offset, i.e. all scalar loads go out together right at the start;
  // the grouped form needs the tile_start search first (a second dependent round trip).
  int pi = 0;
  if (!SINGLE) {
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i)
      if (i < gp->count && bid >= gp->tile_start[i]) pi = i;
    bid -= gp->tile_start[pi];
  }
  const EposPointwiseArgs p = gp->p[pi];
  const int tiles_n = gp->tiles_n[pi];
  const int npad = gp->npad[pi];
  const int tile_n = bid % tiles_n;
  const int tile_m = bid / tiles_n;
  const int m0 = tile_m * BM_, n0 = tile_n * BN_;
  const int M = p.M, N = p.N, K = p.K;
  const int nk = (K + BK - 1) / BK;
  const int cblocks = CONV ? gp->conv_cin[pi] / BK : 1;   // channel blocks per tap
  const int crate = CONV ? gp->conv_rate[pi] : 1;

  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) float*)smem));

  // ---- A pieces: piece = wave*NA + i covers rows 8*piece .. +7, lane -> (row, slot)
  const float* asrc[NA];
  int achunk[NA];
  int apy[CONV ? NA : 1], apx[CONV ? NA : 1];     // CONV: pixel of the lane's row
  unsigned a_dst[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = 8 * (wave * NA + i) + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int m = m0 + r;
    m = m < M ? m : M - 1;
    int64_t row = m;
    if (CONV) {                      // centre tap of output pixel m
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, rem = m - b * hw;
      const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
      apy[i] = yo * p.sub;
      apx[i] = xo * p.sub;
      row = (static_cast<int64_t>(b) * p.Hi + apy[i]) * p.Wi + apx[i];
    } else if (p.sub > 1) {
      // (rare: the stride-2 shortcut convs) keep this a real branch -- if-converted,
      // its two integer divisions per piece sit on every launch's critical path
      asm volatile("" ::: "memory");
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, rem = m - b * hw;
      const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
      row = (static_cast<int64_t>(b) * p.Hi + yo * p.sub) * p.Wi + xo * p.sub;
    }
    asrc[i] = p.A + row * p.lda + c * 4;
    achunk[i] = c * 4;
    a_dst[i] = lds0 + B_BYTES + (wave_u * NA + i) * 1024;
  }
  // ---- W pieces: 1 KiB = 64 columns of one k-group; LDS image [8][BN_][4]
  unsigned wvoff[NW], w_dst[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int piece = wave * NW + i;
    const int q = LAYOUT == 0 ? piece >> 1 : piece, half = LAYOUT == 0 ? piece & 1 : 0;
    wvoff[i] = static_cast<unsigned>((q * npad + half * 64 + lane) * 16);
    w_dst[i] = lds0 + (wave_u * NW + i) * 1024;
  }
  const float* wsb = p.Wp + static_cast<int64_t>(n0) * 4;       // uniform
  const int64_t wstep_tile = static_cast<int64_t>(8) * npad * 4;

  // One LDS-DMA piece of tile kt (pieces 0..NA-1: A rows, NA..5: W k-groups).
  auto issue_piece = [&](int kt, int stage, auto piece_tag, auto tail_tag) {
    constexpr int PIECE = decltype(piece_tag)::value;
    constexpr bool TAIL = decltype(tail_tag)::value;
    const unsigned so = static_cast<unsigned>(stage) * DMA_STAGE_BYTES;
    if constexpr (PIECE < NA) {
      const float* src;
      if constexpr (CONV) {
        const int tap = kt / cblocks, cb = kt - tap * cblocks;       // uniform
        const int ky = tap / 3, dy = (ky - 1) * crate, dx = (tap - ky * 3 - 1) * crate;
        const bool ok = static_cast<unsigned>(apy[PIECE] + dy) < static_cast<unsigned>(p.Hi) &&
                        static_cast<unsigned>(apx[PIECE] + dx) < static_cast<unsigned>(p.Wi);
        src = asrc[PIECE] + ((dy * p.Wi + dx) * p.lda + cb * BK);
        src = ok ? src : g_zero_chunk;
      } else {
        src = asrc[PIECE] + kt * BK;
        if (TAIL) src = (kt * BK + achunk[PIECE] < K) ? src : g_zero_chunk;
      }
      glds16_v(src, a_dst[PIECE] + so);
    } else {
      glds16_s(wvoff[PIECE - NA], wsb + kt * wstep_tile, w_dst[PIECE - NA] + so);
    }
  };
  auto issue = [&](int kt, int stage, auto tail_tag) {
    issue_piece(kt, stage, std::integral_constant<int, 0>{}, tail_tag);
    issue_piece(kt, stage, std::integral_constant<int, 1>{}, tail_tag);
    issue_piece(kt, stage, std::integral_constant<int, 2>{}, tail_tag);
    issue_piece(kt, stage, std::integral_constant<int, 3>{}, tail_tag);
    issue_piece(kt, stage, std::integral_constant<int, 4>{}, tail_tag);
    issue_piece(kt, stage, std::integral_constant<int, 5>{}, tail_tag);
  };

  // ---- fragment addresses (floats from smem); stage offsets are immediates
  int a_off[4];
  {
    const int sw = (l31 >> 1) & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      a_off[g] = B_BYTES / 4 + (wm * 32 + l31) * BK + (((2 * g + h) ^ sw) << 2);
  }
  const int b_off = (h * BN_ + wn * 64 + l31) * 4;
  float4 fa, fb[2];
  auto read_frags = [&](int stage, auto g_tag) {
    constexpr int g = decltype(g_tag)::value;
    const float* s = smem + stage * (DMA_STAGE_BYTES / 4);
    fa = *reinterpret_cast<const float4*>(s + a_off[g]);
    fb[0] = *reinterpret_cast<const float4*>(s + b_off + g * 2 * BN_ * 4);
    fb[1] = *reinterpret_cast<const float4*>(s + b_off + g * 2 * BN_ * 4 + 32 * 4);
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- prologue: tiles 0 and 1 in flight, tile 0 landed + visible -----------
#ifdef EPOS_GEMM_TRACE
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // finer prologue split (first 512 WGs)
  if (t == 0 && blockIdx.x < 512) g_trace_units[blockIdx.x * 32] = wall_clock64();
#endif
  if (nk == 1) {
    issue(0, 0, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    issue(0, 0, std::false_type{});
    if (nk == 2) issue(1, 1, std::true_type{}); else issue(1, 1, std::false_type{});
#ifdef EPOS_GEMM_TRACE
    if (t == 0 && blockIdx.x < 512) g_trace_units[blockIdx.x * 32 + 1] = wall_clock64();
#endif
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  EPOS_TRACE(1);
  read_frags(0, std::integral_constant<int, 0>{});

  // MODE 0: issue tile kt+2 (full)   1: issue tile kt+2 (the last, maybe partial)
  //      2: nothing to issue, tile kt+1 is the last   3: last tile
  auto tile = [&](int kt, int stage, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    const int s2 = stage >= 1 ? stage - 1 : 2;        // (stage + 2) % 3
    const int s1 = stage == 2 ? 0 : stage + 1;
    const int kleft = K - kt * BK;
    // The six LDS-DMA pieces of tile kt+2 are issued ONE PER MFMA PAIR in k-groups
    // 0 and 1: a piece costs ~60 issue cycles, which fits in the shadow of a
    // 64-cycle MFMA but, issued back to back, would drain the matrix pipe.
    auto group = [&](auto g_tag) {
      constexpr int g = decltype(g_tag)::value;
      const float4 ca = fa, cb0 = fb[0], cb1 = fb[1];
      if constexpr (g < 3) {
#ifndef EPOS_ABL_NOREAD
        read_frags(stage, std::integral_constant<int, g + 1>{});
#endif
      } else if constexpr (MODE != 3) {
        // my reads of this stage are complete, my pieces of tile kt+1 have landed
#ifndef EPOS_ABL_NOBAR
        if (MODE <= 1) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
        read_frags(s1, std::integral_constant<int, 0>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 3 || g * 8 < kleft) {
        const float* afp = reinterpret_cast<const float*>(&ca);
        const float* b0p = reinterpret_cast<const float*>(&cb0);
        const float* b1p = reinterpret_cast<const float*>(&cb1);
        auto step = [&](auto s_tag) {
          constexpr int sidx = decltype(s_tag)::value;
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(afp[sidx], b0p[sidx], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(afp[sidx], b1p[sidx], acc[1], 0, 0, 0);
          constexpr int piece = g * 4 + sidx;
#ifdef EPOS_ABL_NODMA
          constexpr bool kIssue = false;
#else
          constexpr bool kIssue = true;
#endif
          if constexpr (kIssue && MODE <= 1 && piece < 6) {
            __builtin_amdgcn_sched_barrier(0);
            issue_piece(kt + 2, s2, std::integral_constant<int, piece>{},
                        std::integral_constant<bool, MODE == 1>{});
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
      }
    };
    group(std::integral_constant<int, 0>{});
    group(std::integral_constant<int, 1>{});
    group(std::integral_constant<int, 2>{});
    group(std::integral_constant<int, 3>{});
  };
  {
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    using M3 = std::integral_constant<int, 3>;
    int kt = 0;
    // steady state, unrolled by the ring length so that every LDS offset is an
    // immediate: tiles whose kt+2 is a full (not the last) tile
    for (; kt + 5 < nk; kt += 3) {
      tile(kt, 0, M0{});
      tile(kt + 1, 1, M0{});
      tile(kt + 2, 2, M0{});
    }
    int stage = 0;                                   // kt is a multiple of 3 here
    for (; kt + 3 < nk; ++kt) {                      // 0..2 more steady tiles
      tile(kt, stage, M0{});
      stage = stage == 2 ? 0 : stage + 1;
    }
    if (kt + 3 == nk) {
      tile(kt, stage, M1{});
      stage = stage == 2 ? 0 : stage + 1;
      ++kt;
    }
    if (kt + 2 == nk) {
      tile(kt, stage, M2{});
      stage = stage == 2 ? 0 : stage + 1;
      ++kt;
    }
    tile(kt, stage, M3{});
  }
  EPOS_TRACE(2);

  // ---- epilogue (as pointwise_gemm_f32<64>) ---------------------------------
  if (vec_epilogue_ok(p, HAS_RES)) {
    __syncthreads();
    float* ws = smem + wave * 32 * EP_ROW;
    vec_epilogue<1, 2, HAS_RES>(ws, acc, p, m0 + wm * 32, n0 + wn * 64, lane);
#ifdef EPOS_GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    EPOS_TRACE(3);
#ifdef EPOS_GEMM_TRACE
    if (t == 0 && blockIdx.x < 8192) g_trace[blockIdx.x * 8 + 6] = clock64();
#endif
    return;
  }
  const bool relu = p.relu != 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
    const int nc = n < N ? n : N - 1;
    const float bias = p.bias ? p.bias[nc] : 0.f;
    const int mb = m0 + wm * 32 + 4 * h;
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
}

template <bool HAS_RES, int LAYOUT, bool CONV, bool SINGLE>
int launch_dma_tt(const GroupedArgs& g, int total, hipStream_t s) {
  static LdsAttrOnce once;
  {
    const int rc = ensure_dynamic_lds(
        once, reinterpret_cast<const void*>(pointwise_gemm_dma_f32<HAS_RES, LAYOUT, CONV, SINGLE>),
        DMA_LDS_BYTES, "hipFuncSetAttribute(pointwise_gemm_dma_f32)");
    if (rc) return rc;
  }
  // 72 KB per workgroup: at most two per CU = two MFMA waves per SIMD
  hipLaunchKernelGGL((pointwise_gemm_dma_f32<HAS_RES, LAYOUT, CONV, SINGLE>), dim3(total),
                     dim3(THREADS), DMA_LDS_BYTES, s, g);
  return launch_status("pointwise_gemm_dma_f32");
}
template <bool HAS_RES, int LAYOUT, bool CONV>
int launch_dma_t(const GroupedArgs& g, int total, hipStream_t s) {
  return g.count == 1 ? launch_dma_tt<HAS_RES, LAYOUT, CONV, true>(g, total, s)
                      : launch_dma_tt<HAS_RES, LAYOUT, CONV, false>(g, total, s);
}

}  // namespace

int launch_grouped_dma(const EposPointwiseArgs* args, int count, hipStream_t s,
                       const int* conv_cin, const int* conv_rate) {

  GroupedArgs g;
  g.count = count;
  int max_n = 0;
  for (int i = 0; i < count; ++i) max_n = args[i].N > max_n ? args[i].N : max_n;
  const bool narrow = max_n <= 64;             // 128 x 64 tiles (LAYOUT 1)
  const int bm = narrow ? 128 : 64, bn = narrow ? 64 : BN;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    g.p[i] = args[i];
    g.npad[i] = static_cast<int>(round_up(args[i].N, BN));
    g.tiles_n[i] = static_cast<int>(ceil_div(args[i].N, bn));
    g.conv_cin[i] = conv_cin ? conv_cin[i] : 0;
    g.conv_rate[i] = conv_rate ? conv_rate[i] : 1;
    g.tile_start[i] = total;
    total += static_cast<int>(ceil_div(args[i].M, bm)) * g.tiles_n[i];
  }
  for (int i = count; i <= MAX_GROUP; ++i) g.tile_start[i] = total;
  const bool res = args[0].R != nullptr;
  if (conv_cin) {
    if (res) {
      set_error("launch_grouped_dma: implicit conv with a residual is not instantiated");
      return EPOS_E_INVALID;
    }
    return narrow ? launch_dma_t<false, 1, true>(g, total, s)
                  : launch_dma_t<false, 0, true>(g, total, s);
  }
  if (narrow)
    return res ? launch_dma_t<true, 1, false>(g, total, s)
               : launch_dma_t<false, 1, false>(g, total, s);
  return res ? launch_dma_t<true, 0, false>(g, total, s)
             : launch_dma_t<false, 0, false>(g, total, s);
}

namespace {
const int registered_dma = (fp32_mfma_ref().dma = &launch_grouped_dma, 0);
}  // namespace
}  // namespace epos

#ifdef EPOS_GEMM_TRACE
extern "C" int epos_debug_read_trace(void* dst, size_t bytes) {
  return static_cast<int>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(epos::g_trace), bytes));
}
extern "C" int epos_debug_read_trace_units(void* dst, size_t bytes) {
  return static_cast<int>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(epos::g_trace_units), bytes));
}
#endif
