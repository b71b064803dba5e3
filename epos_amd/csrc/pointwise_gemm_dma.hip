// LDS-DMA pointwise GEMM kernels for gfx950: the data-parallel default
// (pointwise_gemm_dma_f32) and the persistent stream-K variant on the same ring
// (pointwise_gemm_sk_f32). Tiling, operand layouts and the C-level contract are
// those of pointwise_gemm.hip; DESIGN.md ("The dominant kernel") holds the
// measurements behind every choice below.
#include "pointwise_gemm.h"

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

// ---------------------------------------------------------------------------
// Persistent stream-K variant on top of the LDS-DMA ring. The data-parallel
// kernels give every output tile to one workgroup; with 450 tiles on 512 slots
// (the 48 middle-flow layers) or 600 (N = 1024) the grid is 12-42 % unbalanced and
// every workgroup pays its first-load latency and epilogue in lock step with all
// the others. Here the grid is FIXED at two workgroups per CU (two MFMA waves per
// SIMD = the full-rate regime) and the (tile, K tile) UNITS of all problems of
// the group are cut into equal contiguous ranges, one per workgroup. A worker
// streams its units through ONE continuous three-stage DMA pipeline (the loads of
// unit u+2 are issued while unit u is computed, whatever tile they belong to), so
// a tile change costs an epilogue but no pipeline restart, and epilogues of
// different workers fall at different times and hide behind the co-resident
// workgroup's MFMAs. A range that starts or ends inside a tile gives a partial sum:
//   * a worker whose range starts inside a tile handles that part FIRST, writes
//     the partial accumulators to its slab and raises its flag (it never waits
//     before doing so => no deadlock, whatever the dispatch order);
//   * the worker that owns the tile's first K tile finishes it: it waits for the
//     flags of the following workers, adds their slabs in worker order
//     (deterministic), and runs the normal epilogue.
// Hand-off = slab stores -> vmcnt(0) -> barrier -> one-lane agent-scope release ->
// flag; consumer = relaxed poll -> one agent-scope acquire -> barrier -> plain
// loads (cdna_hip_programming.md section 6, Guideline 16). Flags are reset by the
// consumer, so a launch leaves them zero for the next one; launches that share a
// workspace must be stream-ordered (one workspace per network plan).
// ---------------------------------------------------------------------------
struct SkArgs {
  EposPointwiseArgs p[MAX_GROUP];
  int unit_start[MAX_GROUP + 1];   // prefix sum of tiles * nk
  int tiles_n[MAX_GROUP];
  int npad[MAX_GROUP];
  int nk[MAX_GROUP];
  int count;
  int workers;
  float* slabs;                    // [workers][64 * 128]
  int* flags;                      // [workers + 1] (last = error word)
};

constexpr int SK_BM = 64;
constexpr int SK_SLAB = SK_BM * BN;          // floats

template <bool HAS_RES>
__global__ __launch_bounds__(THREADS) void pointwise_gemm_sk_f32(SkArgs a_) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  (void)a_;
  const SkArgs* __restrict__ gp = (const SkArgs*)__builtin_amdgcn_kernarg_segment_ptr();

  // logical worker id: workers of one XCD are consecutive in unit space
  const int W = gp->workers;
  int L;
  {
    const int raw = blockIdx.x, x = raw & 7, idx = raw >> 3;
    const int q = W >> 3, r = W & 7;
    L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  const int U = gp->unit_start[MAX_GROUP];
  const int u_begin = static_cast<int>(static_cast<int64_t>(L) * U / W);
  const int u_end = static_cast<int>(static_cast<int64_t>(L + 1) * U / W);
  if (u_begin >= u_end) return;               // more workers than units
  EPOS_TRACE(0);
#ifdef EPOS_GEMM_TRACE
  unsigned long long tr_epi = 0, tr_wait = 0, tr_nseg = 0;
  if (t == 0 && blockIdx.x < 8192) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_trace[blockIdx.x * 8 + 4] = (static_cast<unsigned long long>(xcc) << 32) | hw;
  }
#endif
  float* my_slab = gp->slabs + static_cast<int64_t>(L) * SK_SLAB;

  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) float*)smem));
  unsigned a_dst[2], w_dst[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) a_dst[i] = lds0 + DMA_B_BYTES + (wave_u * 2 + i) * 1024;
#pragma unroll
  for (int i = 0; i < 4; ++i) w_dst[i] = lds0 + (wave_u * 4 + i) * 1024;
  int a_off[4];
  {
    const int sw = (l31 >> 1) & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      a_off[g] = DMA_B_BYTES / 4 + (wm * 32 + l31) * BK + (((2 * g + h) ^ sw) << 2);
  }
  const int b_off = (h * BN + wn * 64 + l31) * 4;

  // ---- load context: the tile the DMA front is in ---------------------------
  // Everything a steady-state unit needs lives in registers (no scalar loads in
  // the loop: they share lgkmcnt with the LDS reads). The chunks of a partial last
  // K tile that lie beyond K are fetched from the zero padding of the packed
  // weights (k-group K/4, column 0: zero by the packing contract).
  const float* l_asrc[2];      // row base + chunk of this lane (K tile 0)
  const float* l_atail[2];     // what this lane fetches for the LAST K tile
  int l_achunk[2];
  unsigned l_wvoff[4];
  const float* l_wptr = nullptr;   // uniform: W rows of the CURRENT K tile, column n0
  int64_t l_wstep = 0;
  int l_kt = 0, l_ktl = 0, l_tile_end = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (wave * 2 + i) + (lane >> 3);
    l_achunk[i] = ((lane & 7) ^ ((r >> 1) & 7)) * 4;
  }
  auto decode_load = [&](int v) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i)
      if (i < gp->count && v >= gp->unit_start[i]) pi = i;
    const EposPointwiseArgs p = gp->p[pi];
    const int nk = gp->nk[pi], tiles_n = gp->tiles_n[pi], npad = gp->npad[pi];
    const int ul = v - gp->unit_start[pi];
    const int tile = __builtin_amdgcn_readfirstlane(ul / nk);
    l_kt = ul - tile * nk;
    l_ktl = nk - 1;
    l_tile_end = gp->unit_start[pi] + (tile + 1) * nk;
    const int tile_m = __builtin_amdgcn_readfirstlane(tile / tiles_n);
    const int tile_n = tile - tile_m * tiles_n;
    const int m0 = tile_m * SK_BM, n0 = tile_n * BN;
    const float* zeros = p.Wp + static_cast<int64_t>(p.K / 4) * npad * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 8 * (wave * 2 + i) + (lane >> 3);
      int m = m0 + r;
      m = m < p.M ? m : p.M - 1;
      int64_t row = m;
      if (p.sub > 1) {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw, rem = m - b * hw;
        const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
        row = (static_cast<int64_t>(b) * p.Hi + yo * p.sub) * p.Wi + xo * p.sub;
      }
      l_asrc[i] = p.A + row * p.lda + l_achunk[i];
      l_atail[i] = (l_ktl * BK + l_achunk[i] < p.K) ? l_asrc[i] + l_ktl * BK : zeros;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave * 4 + i, q = piece >> 1, half = piece & 1;
      l_wvoff[i] = static_cast<unsigned>((q * npad + half * 64 + lane) * 16);
    }
    l_wstep = static_cast<int64_t>(8) * npad * 4;
    l_wptr = uniform_ptr(p.Wp + static_cast<int64_t>(n0) * 4 + l_kt * l_wstep);
  };
  // one LDS-DMA piece of the unit the load context points at (K tile l_kt)
  auto issue_piece = [&](int stage, auto piece_tag) {
    constexpr int PIECE = decltype(piece_tag)::value;
    const unsigned so = static_cast<unsigned>(stage) * DMA_STAGE_BYTES;
    if constexpr (PIECE < 2) {
      const float* src = l_asrc[PIECE] + l_kt * BK;
      src = (l_kt == l_ktl) ? l_atail[PIECE] : src;
      glds16_v(src, a_dst[PIECE] + so);
    } else {
      glds16_s(l_wvoff[PIECE - 2], l_wptr, w_dst[PIECE - 2] + so);
    }
  };
  // moves the load context to unit v (called once per unit, before its pieces)
  auto load_seek = [&](int v) {
    if (v == l_tile_end) {
      decode_load(v);
    } else {
      ++l_kt;
      l_wptr += l_wstep;
    }
  };
  auto issue_all = [&](int stage) {
    issue_piece(stage, std::integral_constant<int, 0>{});
    issue_piece(stage, std::integral_constant<int, 1>{});
    issue_piece(stage, std::integral_constant<int, 2>{});
    issue_piece(stage, std::integral_constant<int, 3>{});
    issue_piece(stage, std::integral_constant<int, 4>{});
    issue_piece(stage, std::integral_constant<int, 5>{});
  };

  // ---- compute context: the tile being accumulated --------------------------
  int c_pi = 0, c_tile = 0, c_kt0 = 0, c_seg_end = 0, c_nk = 0, c_m0 = 0, c_n0 = 0;
  auto decode_compute = [&](int u) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i)
      if (i < gp->count && u >= gp->unit_start[i]) pi = i;
    c_pi = pi;
    c_nk = gp->nk[pi];
    const int tiles_n = gp->tiles_n[pi];
    const int ul = u - gp->unit_start[pi];
    c_tile = ul / c_nk;
    c_kt0 = ul - c_tile * c_nk;
    const int tile_end = gp->unit_start[pi] + (c_tile + 1) * c_nk;
    c_seg_end = tile_end < u_end ? tile_end : u_end;
    c_m0 = (c_tile / tiles_n) * SK_BM;
    c_n0 = (c_tile % tiles_n) * BN;
  };

  float4 fa, fb[2];
  auto read_frags = [&](int stage, auto g_tag) {
    constexpr int g = decltype(g_tag)::value;
    const float* s = smem + stage * (DMA_STAGE_BYTES / 4);
    fa = *reinterpret_cast<const float4*>(s + a_off[g]);
    fb[0] = *reinterpret_cast<const float4*>(s + b_off + g * 2 * BN * 4);
    fb[1] = *reinterpret_cast<const float4*>(s + b_off + g * 2 * BN * 4 + 32 * 4);
  };
  f32x16 acc[2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  };
  zero_acc();

  // ---- prologue: units u_begin, u_begin+1 in flight, the first one landed ----
  int u = u_begin;
  decode_load(u);
  issue_all(0);
  if (u + 1 < u_end) {
    load_seek(u + 1);
    issue_all(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  EPOS_TRACE(1);
  read_frags(0, std::integral_constant<int, 0>{});
  decode_compute(u);

  // One unit on ring stage `stage`. MODE 0: unit u+2 exists (its pieces are issued
  // here, one per MFMA pair)   2: u+1 is the worker's last unit   3: u is the last.
  auto unit = [&](int stage, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    const int s2 = stage >= 1 ? stage - 1 : 2;        // (stage + 2) % 3
    const int s1 = stage == 2 ? 0 : stage + 1;
    if (MODE == 0) load_seek(u + 2);
    __builtin_amdgcn_sched_barrier(0);
    auto group = [&](auto g_tag) {
      constexpr int g = decltype(g_tag)::value;
      const float4 ca = fa, cb0 = fb[0], cb1 = fb[1];
      if constexpr (g < 3) {
        read_frags(stage, std::integral_constant<int, g + 1>{});
      } else if constexpr (MODE != 3) {
        // my reads of this stage are complete, my pieces of unit u+1 have landed
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(s1, std::integral_constant<int, 0>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      const float* afp = reinterpret_cast<const float*>(&ca);
      const float* b0p = reinterpret_cast<const float*>(&cb0);
      const float* b1p = reinterpret_cast<const float*>(&cb1);
      auto step = [&](auto s_tag) {
        constexpr int sidx = decltype(s_tag)::value;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(afp[sidx], b0p[sidx], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(afp[sidx], b1p[sidx], acc[1], 0, 0, 0);
        constexpr int piece = g * 4 + sidx;
        if constexpr (MODE == 0 && piece < 6) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(s2, std::integral_constant<int, piece>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
    };
    group(std::integral_constant<int, 0>{});
    group(std::integral_constant<int, 1>{});
    group(std::integral_constant<int, 2>{});
    group(std::integral_constant<int, 3>{});
  };

  int stage = 0;                 // ring stage of unit u
  // ---- the end of a segment (units [.., u] of tile c_tile are accumulated) ----
  auto segment_end = [&]() {
#ifdef EPOS_ABL_NOEPI
    {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[j][r];
      if (sum == 1.2345e-30f) my_slab[t] = sum;     // keeps the MFMAs alive
      return;
    }
#endif
    const EposPointwiseArgs p = gp->p[c_pi];
    const int M = p.M, N = p.N;
    if (c_kt0 != 0) {
      // partial (tail / middle part of a tile): slab + flag, never waits
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          *reinterpret_cast<float4*>(my_slab + (((wave * 2 + j) * 4 + r4) * 64 + lane) * 4) =
              make_float4(acc[j][4 * r4], acc[j][4 * r4 + 1], acc[j][4 * r4 + 2],
                          acc[j][4 * r4 + 3]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(gp->flags + L, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    const int tile_last_unit = gp->unit_start[c_pi] + (c_tile + 1) * c_nk - 1;
    if (c_seg_end - 1 < tile_last_unit) {
      // head of a split tile: add the partials of the following workers in order
      const int l_last = static_cast<int>(
          ((static_cast<int64_t>(tile_last_unit) + 1) * W + U - 1) / U) - 1;
      for (int w = L + 1; w <= l_last; ++w) {
        // workers with an empty unit range (more workers than units) own nothing
        if (static_cast<int64_t>(w) * U / W >= static_cast<int64_t>(w + 1) * U / W) continue;
        if (t == 0) {
          int spins = 0;
#ifdef EPOS_GEMM_TRACE
          const unsigned long long tw0 = wall_clock64();
#endif
          while (__hip_atomic_load(gp->flags + w, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 24)) {          // never hang: flag an error instead
              __hip_atomic_store(gp->flags + W, 1, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
          }
#ifdef EPOS_GEMM_TRACE
          tr_wait += wall_clock64() - tw0;
#endif
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const float* sl = gp->slabs + static_cast<int64_t>(w) * SK_SLAB;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float4 v = *reinterpret_cast<const float4*>(
                sl + (((wave * 2 + j) * 4 + r4) * 64 + lane) * 4);
            acc[j][4 * r4] += v.x; acc[j][4 * r4 + 1] += v.y;
            acc[j][4 * r4 + 2] += v.z; acc[j][4 * r4 + 3] += v.w;
          }
        __syncthreads();
        if (t == 0)
          __hip_atomic_store(gp->flags + w, 0, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      }
    }
    // ---- final epilogue ------------------------------------------------------
    // The ring is busy with the next units, except the stage of the unit just
    // finished. A wave stages through ITS OWN W-piece region of that stage (4 KB:
    // no other wave ever writes there, and the wave's next DMA into it is issued
    // after this epilogue), 8 rows at a time, and streams float4 rows out. Every
    // load (bias, residual) is unconditional from a clamped address and consumed
    // unconditionally, so nothing is pending behind the predicated stores.
    const bool relu = p.relu != 0;
    __syncthreads();        // all waves are done reading the finished stage
    if (vec_epilogue_ok(p, HAS_RES)) {
      const int sdone = stage == 0 ? 2 : stage - 1;       // stage of unit u-1
      float* ws = smem + sdone * (DMA_STAGE_BYTES / 4) + wave * 1024;
      const int m0w = c_m0 + wm * 32, n0w = c_n0 + wn * 64;
      const int c4 = lane & 15, r0 = lane >> 4;
      const int n = n0w + c4 * 4;
      const int ncl = n < N ? n : 0;
      float bias[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nb = n0w + j * 32 + l31;
        bias[j] = p.bias ? p.bias[nb < N ? nb : N - 1] : 0.f;
      }
      float4 rv[8];
      if (HAS_RES) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          int m = m0w + r0 + 4 * i;
          m = m < M ? m : M - 1;
          rv[i] = *reinterpret_cast<const float4*>(p.R + static_cast<int64_t>(m) * p.ldr + ncl);
        }
      }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            ws[(rr + 4 * h) * EP_ROW + j * 32 + l31] = acc[j][4 * ps + rr] + bias[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        float4 v[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          v[i] = *reinterpret_cast<const float4*>(ws + (r0 + 4 * i) * EP_ROW + c4 * 4);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // reads before the next pass
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int m = m0w + 8 * ps + r0 + 4 * i;
          float4 o = v[i];
          if (HAS_RES) {
            const float4 q = rv[2 * ps + i];
            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
          }
          if (relu) o = relu4(o);
          if (m < M && n < N)
            *reinterpret_cast<float4*>(p.C + static_cast<int64_t>(m) * p.ldc + n) = o;
        }
      }
      return;
    }
    // rows not 16-byte aligned (e.g. the 22-channel object head): straight from the
    // accumulator layout, values first (unconditional), predicated stores last
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = c_n0 + wn * 64 + j * 32 + l31;
      const int nc = n < N ? n : N - 1;
      const float bias = p.bias ? p.bias[nc] : 0.f;
      const int mb = c_m0 + wm * 32 + 4 * h;
      float o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = mb + (r & 3) + 8 * (r >> 2);
        m = m < M ? m : M - 1;
        float v = acc[j][r] + bias;
        if (HAS_RES) v += p.R[static_cast<int64_t>(m) * p.ldr + nc];
        o[r] = relu ? fmaxf(v, 0.f) : v;
      }
      asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]),
                   "+v"(o[5]), "+v"(o[6]), "+v"(o[7]));
      asm volatile("" : "+v"(o[8]), "+v"(o[9]), "+v"(o[10]), "+v"(o[11]), "+v"(o[12]),
                   "+v"(o[13]), "+v"(o[14]), "+v"(o[15]));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        if (m < M && n < N) p.C[static_cast<int64_t>(m) * p.ldc + n] = o[r];
      }
    }
  };

  for (;;) {
    // steady units of the current segment: one plain loop, accumulators in place
    int n0 = (c_seg_end < u_end - 2 ? c_seg_end : u_end - 2) - u;
    for (; n0 > 0; --n0) {
      unit(stage, std::integral_constant<int, 0>{});
      stage = stage == 2 ? 0 : stage + 1;
      ++u;
#ifdef EPOS_GEMM_TRACE
      if (t == 0 && blockIdx.x < 512 && u - u_begin < 32)
        g_trace_units[blockIdx.x * 32 + (u - u_begin)] = wall_clock64();
#endif
    }
    while (u < c_seg_end) {                       // the worker's last two units
      if (u + 1 < u_end) unit(stage, std::integral_constant<int, 2>{});
      else unit(stage, std::integral_constant<int, 3>{});
      stage = stage == 2 ? 0 : stage + 1;
      ++u;
    }
#ifdef EPOS_GEMM_TRACE
    const unsigned long long tr0 = wall_clock64();
#endif
    segment_end();
#ifdef EPOS_GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr_epi += wall_clock64() - tr0;
    ++tr_nseg;
#endif
    if (u == u_end) break;
    // The epilogue's stores are NOT drained: loads (LDS-DMA included) return in
    // order among themselves, so "at most 6 outstanding" still implies that the
    // pieces of unit u+1 have landed (if one of them were pending, the 6 younger
    // pieces of u+2 would be too => more than 6). Stores in flight can only make
    // the counted wait longer, never shorter. Every compiler-visible load of the
    // epilogue is consumed inside it, so hipcc has nothing pending either.
#ifdef EPOS_SK_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
#endif
    zero_acc();
    decode_compute(u);
  }
#ifdef EPOS_GEMM_TRACE
  EPOS_TRACE(3);
  if (t == 0 && blockIdx.x < 8192) {
    g_trace[blockIdx.x * 8 + 2] = tr_epi;
    g_trace[blockIdx.x * 8 + 5] = tr_wait;
    g_trace[blockIdx.x * 8 + 6] = tr_nseg;
    g_trace[blockIdx.x * 8 + 7] = static_cast<unsigned long long>(u_end - u_begin);
  }
#endif
}

template <bool HAS_RES, int LAYOUT, bool CONV, bool SINGLE>
int launch_dma_tt(const GroupedArgs& g, int total, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(pointwise_gemm_dma_f32<HAS_RES, LAYOUT, CONV, SINGLE>),
        hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS_BYTES);
    attr_set = true;
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

constexpr int SK_WORKERS = 512;             // two workgroups per CU on 256 CUs

template <bool HAS_RES>
int launch_sk_t(const SkArgs& a, hipStream_t s) {
  // 72 KB ring per workgroup: exactly two workgroups per CU
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(pointwise_gemm_sk_f32<HAS_RES>),
        hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((pointwise_gemm_sk_f32<HAS_RES>), dim3(a.workers),
                     dim3(THREADS), DMA_LDS_BYTES, s, a);
  return launch_status("pointwise_gemm_sk_f32");
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

int launch_grouped_sk(const EposPointwiseArgs* args, int count, void* workspace,
                      hipStream_t s) {
  SkArgs a;
  a.count = count;
  int64_t units = 0;
  for (int i = 0; i < count; ++i) {
    a.p[i] = args[i];
    a.npad[i] = static_cast<int>(round_up(args[i].N, BN));
    a.tiles_n[i] = a.npad[i] / BN;
    a.nk[i] = static_cast<int>(ceil_div(args[i].K, BK));
    a.unit_start[i] = static_cast<int>(units);
    units += ceil_div(args[i].M, SK_BM) * a.tiles_n[i] * a.nk[i];
  }
  for (int i = count; i <= MAX_GROUP; ++i) a.unit_start[i] = static_cast<int>(units);
  a.workers = SK_WORKERS;
  a.slabs = static_cast<float*>(workspace);
  a.flags = reinterpret_cast<int*>(static_cast<char*>(workspace) +
                                   sizeof(float) * SK_SLAB * SK_WORKERS);
  return args[0].R != nullptr ? launch_sk_t<true>(a, s) : launch_sk_t<false>(a, s);
}

int64_t sk_workspace_bytes() {
  return static_cast<int64_t>(sizeof(float)) * SK_SLAB * SK_WORKERS + 4096;
}

}  // namespace epos

#ifdef EPOS_GEMM_TRACE
extern "C" int epos_debug_read_trace(void* dst, size_t bytes) {
  return static_cast<int>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(epos::g_trace), bytes));
}
extern "C" int epos_debug_read_trace_units(void* dst, size_t bytes) {
  return static_cast<int>(hipMemcpyFromSymbol(dst, HIP_SYMBOL(epos::g_trace_units), bytes));
}
#endif
