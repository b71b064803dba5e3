// fp32 pointwise GEMM on the bf16 matrix pipe of gfx950 ("split" kernel).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate, and the GEMMs are 95 % of
// the network's arithmetic. Here every fp32 operand is cut EXACTLY into three bf16
// pieces -- x = hi + mid + lo, 8 + 8 + 8 significand bits, taken by truncation so that
// both residuals are exact fp32 subtractions -- and the product a*b is assembled from
// six piece products with v_mfma_f32_32x32x16_bf16, accumulated in fp32:
//     a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)  [+ am*bl + al*bm + al*bl]
// The three dropped terms are below 2^-23 |a*b|, i.e. below the rounding of a single
// fp32 multiply-add; the piece products themselves are exact (8 x 8 bits) and each MFMA
// adds 16 of them with one rounding, so the result is NOT less accurate than the fp32
// MFMA (or an fmaf chain): measured against fp64 the rms error is equal with one
// accumulator and ~3x smaller with the correction terms kept in their own accumulator
// (tools/split_proto/split_acc.hip, tests/test_gpu_layers.py). Six bf16 MFMAs of 32
// cycles replace eight fp32 MFMAs of 64 for the same 32x32x16 block: 2.67x less matrix
// pipe time, and the bf16 MFMA does not share the VALU the way the fp32 one does.
//
// Weights are split and laid out in MFMA fragment order once on the host
// (epos_pack_pointwise_weights_split); activations stay fp32 in HBM and are split in
// registers after the fragment read (11 VALU ops per pair of values).
//
// Tile 128 x 128 per 256-thread workgroup, waves 4 x 1: every wave owns 32 rows and
// all 128 columns (4 blocks x 2 accumulators), so each A value is split by exactly one
// wave; K step 16 per stage, FOUR-stage LDS-DMA ring of 20 KB (A 128 rows x 64 B fp32,
// XOR-swizzled through the per-lane source address; W 12 KB lane-linear fragments) =
// 80 KB, two workgroups per CU. Small grids take 64 x 128 tiles (waves 2 x 2, 32 x 64
// each, 64 KB ring). Tile kt+3 is issued while tile kt is computed; one raw s_barrier
// per stage as in pointwise_gemm_dma_f32.
#include <string.h>

#include "pointwise_gemm.h"

namespace epos {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) float g_zero_chunk_sp[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int SP_BN = 128, SP_BK = 16;
constexpr int SP_W_BYTES = SP_BK * SP_BN * 6;        // 12288: 4 col blocks x 3 pieces x 1 KB
constexpr int SP_NST = 4;
// CB = 32-column blocks per wave: CB 4 -> waves 4 x 1, 128 x 128 tile (A 8 KB per stage,
// 80 KB ring); CB 2 -> waves 2 x 2, 64 x 128 tile (A 4 KB, 64 KB ring) for grids that
// would otherwise leave the CUs with one workgroup (= one wave per SIMD) each.
constexpr int sp_stage_bytes(int cb) { return SP_W_BYTES + 32 * cb * SP_BK * 4; }
constexpr int sp_lds_bytes(int cb) { return SP_NST * sp_stage_bytes(cb); }
constexpr int sp_ep_row(int cb) { return cb * 32 + 4; }     // floats per staged row

__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) {   // {b.hi, a.hi}
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}

// eight fp32 values (k = 0..7 of one row) -> three packed bf16x8 pieces
__device__ __forceinline__ void split8(const float4 x0, const float4 x1, u32x4& hi,
                                       u32x4& mid, u32x4& lo) {
  const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  unsigned hb[8], mb[8], lb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hb[j] = __float_as_uint(x[j]);
    const float r1 = x[j] - __uint_as_float(hb[j] & 0xffff0000u);   // exact
    mb[j] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(mb[j] & 0xffff0000u);     // exact, <= 8 bits
    lb[j] = __float_as_uint(r2);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = pack_hi16(hb[2 * j], hb[2 * j + 1]);
    mid[j] = pack_hi16(mb[2 * j], mb[2 * j + 1]);
    lo[j] = pack_hi16(lb[2 * j], lb[2 * j + 1]);
  }
}

__device__ __forceinline__ void mfma_bf16(const u32x4& a, const u32x4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                              __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// CONV: implicit GEMM of a dense 3x3 conv with Cin % 16 == 0, as in
// pointwise_gemm_dma_f32: K step kt is channel block kt % (Cin/16) of tap kt / (Cin/16),
// the A rows are the input pixels (y*stride + dy*rate, x*stride + dx*rate), taps outside
// the image come from a zero block (a per-lane source select).
template <bool HAS_RES, bool SINGLE, bool TWO_ACC, int CB, bool CONV>
__global__ __launch_bounds__(THREADS, 2) void pointwise_gemm_split_f32(GroupedArgs ga_) {
  constexpr int WN = 4 / CB;                 // waves along N (1 or 2); CB along M
  constexpr int SP_BM = 32 * CB;
  constexpr int SP_STAGE = sp_stage_bytes(CB);
  constexpr int RB = CB / 2;                 // A pieces (16 rows) per wave and stage
  constexpr int NP = RB + 3;                 // LDS-DMA pieces per wave and stage
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;

  (void)ga_;
  const GroupedArgs* __restrict__ gp =
      (const GroupedArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  int bid;
  {   // workgroups of one XCD (blockIdx % 8) take a contiguous range of tiles
    const int total = gp->tile_start[MAX_GROUP];
    const int raw = blockIdx.x, x = raw & 7, idx = raw >> 3;
    const int q = total >> 3, r = total & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
  }
  int pi = 0;
  if (!SINGLE) {
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i)
      if (i < gp->count && bid >= gp->tile_start[i]) pi = i;
    bid -= gp->tile_start[pi];
  }
  const EposPointwiseArgs p = gp->p[pi];
  const int tiles_n = gp->tiles_n[pi];
  const int M = p.M, N = p.N, K = p.K;
  // Tile order inside an XCD's range: column fastest, but for wide problems (more than 8
  // column tiles: the 4032- and 1344-channel heads, N = 1536 / 2048) in bands of 8 column
  // tiles with all row tiles of a band before the next band, so that the workgroups in
  // flight on an XCD share 8 tiles' weights (K = 256: 1.5 MB, L2-resident) instead of
  // sweeping all of them (6 MB for the 4032-channel head) once per row tile.
  int tile_m, tile_n;
  if (tiles_n <= 8) {
    tile_n = bid % tiles_n;
    tile_m = bid / tiles_n;
  } else {
    const int tiles_m = (M + SP_BM - 1) / SP_BM;
    const int per_band = tiles_m * 8;
    const int band = bid / per_band;
    const int rem = bid - band * per_band;
    const int left = tiles_n - band * 8;
    const int bw = left < 8 ? left : 8;
    tile_m = rem / bw;
    tile_n = band * 8 + (rem - tile_m * bw);
  }
  const int m0 = tile_m * SP_BM, n0 = tile_n * SP_BN;
  const int nks = (K + SP_BK - 1) / SP_BK;
  const int cblocks = CONV ? gp->conv_cin[pi] / SP_BK : 1;   // channel blocks per tap
  const int crate = CONV ? gp->conv_rate[pi] : 1;

  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) float*)smem));

  // ---- A pieces (1 KB = 16 rows x 64 B): piece = wave*2 + i, lane -> (row, slot);
  //      slot s of row r holds chunk s ^ ((r >> 2) & 3)
  const float* asrc[RB];
  unsigned avoff[RB];        // the same as a 32-bit byte offset from p.A (scalar-base form)
  int achunk[RB];
  int apy[CONV ? RB : 1], apx[CONV ? RB : 1];     // CONV: pixel of the lane's row
  unsigned a_dst[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int r = 16 * (wave * RB + i) + (lane >> 2);
    const int c = (lane & 3) ^ ((r >> 2) & 3);
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
      asm volatile("" ::: "memory");        // keep the divisions off the common path
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, rem = m - b * hw;
      const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
      row = (static_cast<int64_t>(b) * p.Hi + yo * p.sub) * p.Wi + xo * p.sub;
    }
    asrc[i] = p.A + row * p.lda + c * 4;
    avoff[i] = static_cast<unsigned>((row * p.lda + c * 4) * 4);   // bytes from p.A
    achunk[i] = c * 4;
    a_dst[i] = lds0 + SP_W_BYTES + (wave_u * RB + i) * 1024;
  }
  // ---- W pieces: the 12 KB stage image is contiguous in the packed buffer
  unsigned wvoff[3], w_dst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    wvoff[i] = static_cast<unsigned>(((wave * 3 + i) * 64 + lane) * 16);
    w_dst[i] = lds0 + (wave_u * 3 + i) * 1024;
  }
  const float* abase = uniform_ptr(p.A);
  const float* wsb = uniform_ptr(reinterpret_cast<const float*>(
      static_cast<const char*>(p.Ws) + static_cast<int64_t>(tile_n) * nks * SP_W_BYTES));

  auto issue_piece = [&](int kt, int stage, auto piece_tag, auto tail_tag) {
#ifdef EPOS_SPLIT_ABL_DMAHOT
    kt = kt & 1;        // ablation: always the same two K steps (cache-hot sources)
#endif
    constexpr int PIECE = decltype(piece_tag)::value;
    constexpr bool TAIL = decltype(tail_tag)::value;
    const unsigned so = static_cast<unsigned>(stage) * SP_STAGE;
    if constexpr (PIECE < RB) {
      const float* src;
      if constexpr (CONV) {
        const int tap = kt / cblocks, cb = kt - tap * cblocks;       // uniform
        const int ky = tap / 3, dy = (ky - 1) * crate, dx = (tap - ky * 3 - 1) * crate;
        const bool ok = static_cast<unsigned>(apy[PIECE] + dy) < static_cast<unsigned>(p.Hi) &&
                        static_cast<unsigned>(apx[PIECE] + dx) < static_cast<unsigned>(p.Wi);
        src = asrc[PIECE] + ((dy * p.Wi + dx) * p.lda + cb * SP_BK);
        src = ok ? src : g_zero_chunk_sp;
      } else if constexpr (!TAIL) {
        // full K step of a 1x1 conv: scalar base + 32-bit lane offset. The LDS-DMA
        // instruction with a 64-bit per-lane address costs the issuing wave noticeably
        // more (tools/mfma_peak/mfma_bf16.hip modes 4 / 5: 5 pieces per 24 MFMAs, two
        // waves per SIMD, 1657 -> 1732 TFLOP/s bf16); the pieces that may need the zero
        // block (last partial K step, implicit-conv taps) keep the per-lane pointer.
        const float* ab = abase + (kt * SP_BK - PIECE * 256);      // uniform
#ifdef EPOS_SPLIT_M0_EACH
        glds16_s_m0(avoff[PIECE], ab + PIECE * 256, a_dst[PIECE] + so);
#else
        if constexpr (PIECE == 0) glds16_s_m0(avoff[0], ab, a_dst[0] + so);
        else glds16_s_off<PIECE * 1024>(avoff[PIECE], ab);
#endif
        return;
      } else {
        src = asrc[PIECE] + kt * SP_BK;
        if (TAIL) src = (kt * SP_BK + achunk[PIECE] < K) ? src : g_zero_chunk_sp;
      }
      // pieces of one kind are 1 KB apart in LDS: one M0 write for the first, the
      // instruction offset (added to both addresses) for the others
#ifdef EPOS_SPLIT_M0_EACH
      glds16_v_m0(src, a_dst[PIECE] + so);
#else
      if constexpr (PIECE == 0) glds16_v_m0(src, a_dst[0] + so);
      else glds16_v_off<PIECE * 1024>(src - PIECE * 256);
#endif
    } else {
      const float* wb = wsb + static_cast<int64_t>(kt) * (SP_W_BYTES / 4);
#ifdef EPOS_SPLIT_M0_EACH
      glds16_s_m0(wvoff[PIECE - RB], wb, w_dst[PIECE - RB] + so);
#else
      if constexpr (PIECE == RB) glds16_s_m0(wvoff[0], wb, w_dst[0] + so);
      else glds16_s_off<(PIECE - RB) * 1024>(wvoff[0], wb);
#endif
    }
  };
  auto issue = [&](int kt, int stage) {
    issue_piece(kt, stage, std::integral_constant<int, 0>{}, std::true_type{});
    issue_piece(kt, stage, std::integral_constant<int, 1>{}, std::true_type{});
    issue_piece(kt, stage, std::integral_constant<int, 2>{}, std::true_type{});
    issue_piece(kt, stage, std::integral_constant<int, 3>{}, std::true_type{});
    if constexpr (NP > 4)
      issue_piece(kt, stage, std::integral_constant<int, 4>{}, std::true_type{});
  };

  // ---- fragment addresses (float index from the stage base)
  int a_off[2];
  {
    const int sw = (l31 >> 2) & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      a_off[j] = SP_W_BYTES / 4 + (wm * 32 + l31) * SP_BK + (((2 * h + j) ^ sw) << 2);
  }
  const int b_off = (wn * CB * 3 * 64 + lane) * 4;     // + (cb*3 + piece) * 256 floats

  float4 xa[2];             // raw fp32 A fragments of the NEXT stage to compute
  u32x4 bp[CB][3];          // pre-split W fragments
  auto read_a = [&](int stage) {
    const float* s = smem + stage * (SP_STAGE / 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) xa[j] = *reinterpret_cast<const float4*>(s + a_off[j]);
  };
  auto read_b = [&](int stage, auto cb_tag) {
    constexpr int cb = decltype(cb_tag)::value;
    const float* s = smem + stage * (SP_STAGE / 4);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
      bp[cb][pc] = *reinterpret_cast<const u32x4*>(s + b_off + (cb * 3 + pc) * 256);
  };

  f32x16 acc[CB], acc2[TWO_ACC ? CB : 1];
#pragma unroll
  for (int j = 0; j < CB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  if (TWO_ACC) {
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
  }

  // ---- prologue: up to three tiles in flight, tile 0 landed + visible
  issue(0, 0);
  if (nks > 1) issue(1, 1);
  if (nks > 2) issue(2, 2);
  if (nks > 2) {
    if (NP == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else if (nks > 1) {
    if (NP == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_a(0);
  read_b(0, std::integral_constant<int, 0>{});
  read_b(0, std::integral_constant<int, 1>{});
  if constexpr (CB == 4) {
    read_b(0, std::integral_constant<int, 2>{});
    read_b(0, std::integral_constant<int, 3>{});
  }

  // bf16 pieces of this wave's A fragment for the stage being computed; the next
  // stage's are split in the middle of the current one (between its MFMAs)
  u32x4 ah, am, al;
  auto split_xa = [&](u32x4& hi, u32x4& mid, u32x4& lo) {
#ifdef EPOS_SPLIT_ABL_NOSPLIT
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* x0 = reinterpret_cast<const float*>(&xa[0]);
      const float* x1 = reinterpret_cast<const float*>(&xa[1]);
      hi[j] = __float_as_uint(x0[j]); mid[j] = __float_as_uint(x1[j]);
      lo[j] = hi[j] ^ mid[j];
    }
#else
    split8(xa[0], xa[1], hi, mid, lo);
#endif
  };
  split_xa(ah, am, al);
  // MODE 0: issue tile kt+3 (full)  1: issue tile kt+3 (the last, maybe partial)
  //      2: kt+2 is the last tile   3: kt+1 is the last tile   4: last tile
  // LIVE: column blocks of this wave that hold any column < N (CB, or CB - 1 for the
  // last column tile of e.g. N = 728 with CB 4: with waves 4 x 1 every wave of the
  // workgroup then skips the same quarter of its MFMAs)
  auto tile = [&](int kt, int stage, auto mode_tag, auto live_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr int LIVE = decltype(live_tag)::value;
    const int s3 = (stage + 3) & 3, s1 = (stage + 1) & 3;
    f32x16* corr = TWO_ACC ? acc2 : acc;
    // one column block: small terms first, into the correction accumulator
    // next stage's pieces, produced value by value between the MFMAs of the second half
    u32x4 nh, nm, nl;
    unsigned hb[8], mb[8], lb[8];
    auto split_val = [&](auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
      const float x = reinterpret_cast<const float*>(&xa[j >> 2])[j & 3];
#ifdef EPOS_SPLIT_ABL_NOSPLIT
      hb[j] = __float_as_uint(x); mb[j] = hb[j] ^ 0x3f80u; lb[j] = hb[j] ^ 0x40u;
#else
      hb[j] = __float_as_uint(x);
      const float r1 = x - __uint_as_float(hb[j] & 0xffff0000u);
      mb[j] = __float_as_uint(r1);
      const float r2 = r1 - __uint_as_float(mb[j] & 0xffff0000u);
      lb[j] = __float_as_uint(r2);
#endif
      if constexpr (j & 1) {
        nh[j >> 1] = pack_hi16(hb[j - 1], hb[j]);
        nm[j >> 1] = pack_hi16(mb[j - 1], mb[j]);
        nl[j >> 1] = pack_hi16(lb[j - 1], lb[j]);
      }
    };
    auto block = [&](auto cb_tag, auto dma_tag, auto slot_tag) {
      constexpr int cb = decltype(cb_tag)::value;
      constexpr int DMA0 = decltype(dma_tag)::value;    // first piece to issue, -1: none
      constexpr int SLOT0 = decltype(slot_tag)::value;  // second half: MFMA slot of n = 0, -1: none
      const u32x4 bh = bp[cb][0], bm = bp[cb][1], bl = bp[cb][2];
      auto one = [&](const u32x4& a, const u32x4& b, f32x16& c, auto n_tag) {
        constexpr int n = decltype(n_tag)::value;       // 0..5 within the block
        mfma_bf16(a, b, c);
#ifdef EPOS_SPLIT_ABL_NODMA
        constexpr bool kIssue = false;
#else
        constexpr bool kIssue = true;
#endif
        // the LDS-DMA pieces of tile kt+3 go out one at a time between MFMAs: after
        // every second MFMA of column blocks 0 and 1 (CB 4), after each of the first
        // four MFMAs of block 0 (CB 2)
        constexpr int piece = DMA0 < 0 ? -1 : CB == 4 ? ((n & 1) ? DMA0 + n / 2 : -1) : DMA0 + n;
        if constexpr (kIssue && piece >= 0 && piece < NP) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(kt + 3, s3, std::integral_constant<int, piece>{},
                      std::integral_constant<bool, MODE == 1>{});
          __builtin_amdgcn_sched_barrier(0);
        }
        // second half: the next stage's A values are split one (CB 4) or two (CB 2) per
        // MFMA, pinned between the MFMAs (the bf16 MFMA leaves the vector ALU free);
        // the first slots cover the LDS latency of the fragment just requested
        if constexpr (SLOT0 >= 0 && MODE != 4) {
          constexpr int slot = SLOT0 + n;
          constexpr bool kOnePerSlot = CB == 4 && LIVE == 4;     // 12 MFMAs, else 6
          constexpr int first = kOnePerSlot ? slot - 2 : 2 * (slot - 1);
          constexpr int cnt = kOnePerSlot ? 1 : 2;
          if constexpr (first >= 0 && first < 8) {
            __builtin_amdgcn_sched_barrier(0);
            split_val(std::integral_constant<int, first>{});
            if constexpr (cnt == 2) split_val(std::integral_constant<int, first + 1>{});
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      one(al, bh, corr[cb], std::integral_constant<int, 0>{});
      one(ah, bl, corr[cb], std::integral_constant<int, 1>{});
      one(am, bm, corr[cb], std::integral_constant<int, 2>{});
      one(am, bh, corr[cb], std::integral_constant<int, 3>{});
      one(ah, bm, corr[cb], std::integral_constant<int, 4>{});
      one(ah, bh, acc[cb], std::integral_constant<int, 5>{});
    };
    using NoDma = std::integral_constant<int, -1>;
    constexpr bool ISSUE = MODE <= 1;
    // Two column blocks interleaved (CB 4): the same six piece products per block in the
    // same order per accumulator (bit-identical sums), but consecutive MFMAs alternate
    // between the two blocks' accumulators. An MFMA that accumulates into the register
    // block its predecessor writes only issues back to back when NOTHING sits between
    // them; the DMA pieces and the operand split pinned between such a pair cost ~40+
    // cycles each (MI355X_MICROARCH.md, "extra issue slot between two MFMAs on the same
    // accumulator"), between MFMAs on different accumulators they hide.
    auto pair = [&](auto ca_tag, auto cb_tag, auto dma_tag, auto slot_tag) {
      constexpr int ca = decltype(ca_tag)::value, cbb = decltype(cb_tag)::value;
      constexpr int DMA0 = decltype(dma_tag)::value;     // 0: issue pieces 0..NP-1, -1: none
      constexpr int SLOT0 = decltype(slot_tag)::value;   // 0: split the next A values, -1: no
      auto step = [&](const u32x4& a, const u32x4& b, f32x16& c, auto s_tag) {
        constexpr int s = decltype(s_tag)::value;        // 0..11 within the pair
        mfma_bf16(a, b, c);
#ifdef EPOS_SPLIT_ABL_NODMA
        constexpr bool kIssue = false;
#else
        constexpr bool kIssue = true;
#endif
        constexpr int piece = (DMA0 < 0 || !(s & 1)) ? -1 : DMA0 + s / 2;
        if constexpr (kIssue && piece >= 0 && piece < NP) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(kt + 3, s3, std::integral_constant<int, piece>{},
                      std::integral_constant<bool, MODE == 1>{});
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SLOT0 >= 0 && MODE != 4) {
          constexpr int first = SLOT0 + s - 2;           // one value per MFMA, slots 2..9
          if constexpr (first >= 0 && first < 8) {
            __builtin_amdgcn_sched_barrier(0);
            split_val(std::integral_constant<int, first>{});
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      step(al, bp[ca][0], corr[ca], std::integral_constant<int, 0>{});
      step(al, bp[cbb][0], corr[cbb], std::integral_constant<int, 1>{});
      step(ah, bp[ca][2], corr[ca], std::integral_constant<int, 2>{});
      step(ah, bp[cbb][2], corr[cbb], std::integral_constant<int, 3>{});
      step(am, bp[ca][1], corr[ca], std::integral_constant<int, 4>{});
      step(am, bp[cbb][1], corr[cbb], std::integral_constant<int, 5>{});
      step(am, bp[ca][0], corr[ca], std::integral_constant<int, 6>{});
      step(am, bp[cbb][0], corr[cbb], std::integral_constant<int, 7>{});
      step(ah, bp[ca][1], corr[ca], std::integral_constant<int, 8>{});
      step(ah, bp[cbb][1], corr[cbb], std::integral_constant<int, 9>{});
      step(ah, bp[ca][0], acc[ca], std::integral_constant<int, 10>{});
      step(ah, bp[cbb][0], acc[cbb], std::integral_constant<int, 11>{});
    };
#ifndef EPOS_SPLIT_NOILV
    constexpr bool kPair = CB == 4;
#else
    constexpr bool kPair = false;
#endif
    // first half of the column blocks (+ the DMA pieces), barrier, second half
    if constexpr (kPair) {
      pair(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{},
           std::integral_constant<int, ISSUE ? 0 : -1>{}, NoDma{});
    } else if constexpr (CB == 4) {
      block(std::integral_constant<int, 0>{}, std::integral_constant<int, ISSUE ? 0 : -1>{}, NoDma{});
      block(std::integral_constant<int, 1>{}, std::integral_constant<int, ISSUE ? 3 : -1>{}, NoDma{});
    } else {
      block(std::integral_constant<int, 0>{}, std::integral_constant<int, ISSUE ? 0 : -1>{}, NoDma{});
    }
    if constexpr (MODE != 4) {
      // my reads of this stage are complete (fragments are in registers); my pieces
      // of tile kt+1 have landed once at most the later tiles' pieces are outstanding
#ifndef EPOS_SPLIT_ABL_NOBAR
      if (MODE <= 1) {
        if (NP == 5) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      } else if (MODE == 2) {
        if (NP == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      }
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#endif
#ifndef EPOS_SPLIT_ABL_NOREAD
      read_a(s1);
      read_b(s1, std::integral_constant<int, 0>{});
      if constexpr (CB == 4) read_b(s1, std::integral_constant<int, 1>{});
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (kPair && LIVE == 4) {
      pair(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, NoDma{},
           std::integral_constant<int, 0>{});
    } else if constexpr (CB == 4) {
      block(std::integral_constant<int, 2>{}, NoDma{}, std::integral_constant<int, 0>{});
      if constexpr (LIVE == 4)
        block(std::integral_constant<int, 3>{}, NoDma{}, std::integral_constant<int, 6>{});
    } else {
      block(std::integral_constant<int, 1>{}, NoDma{}, std::integral_constant<int, 0>{});
    }
#ifndef EPOS_SPLIT_ABL_NOREAD
    if constexpr (MODE != 4) {
      if constexpr (CB == 4) {
        read_b(s1, std::integral_constant<int, 2>{});
        if constexpr (LIVE == 4) read_b(s1, std::integral_constant<int, 3>{});
      } else {
        read_b(s1, std::integral_constant<int, 1>{});
      }
    }
#endif
    if constexpr (MODE != 4) { ah = nh; am = nm; al = nl; }
  };
  auto k_loop = [&](auto live_tag) {
    using LV = decltype(live_tag);
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    using M3 = std::integral_constant<int, 3>;
    using M4 = std::integral_constant<int, 4>;
    int kt = 0;
    for (; kt + 7 < nks; kt += 4) {        // every LDS offset an immediate
      tile(kt, 0, M0{}, LV{});
      tile(kt + 1, 1, M0{}, LV{});
      tile(kt + 2, 2, M0{}, LV{});
      tile(kt + 3, 3, M0{}, LV{});
    }
    int stage = 0;                          // kt is a multiple of 4 here
    for (; kt + 4 < nks; ++kt) {
      tile(kt, stage, M0{}, LV{});
      stage = (stage + 1) & 3;
    }
    if (kt + 4 == nks) { tile(kt, stage, M1{}, LV{}); stage = (stage + 1) & 3; ++kt; }
    if (kt + 3 == nks) { tile(kt, stage, M2{}, LV{}); stage = (stage + 1) & 3; ++kt; }
    if (kt + 2 == nks) { tile(kt, stage, M3{}, LV{}); stage = (stage + 1) & 3; ++kt; }
    tile(kt, stage, M4{}, LV{});
  };
  if (CB == 4 && n0 + 96 >= N) k_loop(std::integral_constant<int, 3>{});   // uniform
  else k_loop(std::integral_constant<int, CB>{});
  if (TWO_ACC) {
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] += acc2[j][r];
  }

  // ---- epilogue --------------------------------------------------------------
  if (vec_epilogue_ok(p, HAS_RES)) {
    __syncthreads();
    float* ws = smem + wave * 32 * sp_ep_row(CB);
    vec_epilogue<1, CB, HAS_RES, sp_ep_row(CB)>(ws, acc, p, m0 + wm * 32,
                                                 n0 + wn * CB * 32, lane);
    return;
  }
  const bool relu = p.relu != 0;
  constexpr int i = 0;
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int n = n0 + wn * CB * 32 + j * 32 + l31;
      const int nc = n < N ? n : N - 1;
      const float bias = p.bias ? p.bias[nc] : 0.f;
      const int mb = m0 + wm * 32 + i * 32 + 4 * h;
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
        float v = acc[i * CB + j][r] + bias;
        if (HAS_RES) v += rv[r];
        if (relu) v = fmaxf(v, 0.f);
        if (m < M && n < N) p.C[static_cast<int64_t>(m) * p.ldc + n] = v;
      }
    }
}

template <bool HAS_RES, bool SINGLE, int CB, bool TWO_ACC, bool CONV = false>
int launch_split_tt(const GroupedArgs& g, int total, hipStream_t s) {
  auto kern = pointwise_gemm_split_f32<HAS_RES, SINGLE, TWO_ACC, CB, CONV>;
  static LdsAttrOnce once;
  {
    const int rc = ensure_dynamic_lds(once, reinterpret_cast<const void*>(kern), sp_lds_bytes(CB),
                                      "hipFuncSetAttribute(pointwise_gemm_split_f32)");
    if (rc) return rc;
  }
  // 80 / 64 KB per workgroup: at most two per CU = two MFMA waves per SIMD
  hipLaunchKernelGGL(kern, dim3(total), dim3(THREADS), sp_lds_bytes(CB), s, g);
  return launch_status("pointwise_gemm_split_f32");
}

template <int CB>
int launch_split_rb(const EposPointwiseArgs* args, int count, hipStream_t s,
                    const int* conv_cin, const int* conv_rate) {
  GroupedArgs g = {};
  g.count = count;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    g.p[i] = args[i];
    g.tile_start[i] = total;
    g.tiles_n[i] = static_cast<int>(ceil_div(args[i].N, SP_BN));
    g.npad[i] = g.tiles_n[i] * SP_BN;
    g.conv_cin[i] = conv_cin ? conv_cin[i] : 0;
    g.conv_rate[i] = conv_rate ? conv_rate[i] : 1;
    total += static_cast<int>(ceil_div(args[i].M, 32 * CB)) * g.tiles_n[i];
  }
  for (int i = count; i <= MAX_GROUP; ++i) g.tile_start[i] = total;
  const bool res = args[0].R != nullptr;
  const bool single = count == 1;
  if (conv_cin) {                       // one problem, no residual (checked by the caller)
    return launch_split_tt<false, true, CB, true, true>(g, total, s);
  }
  // EPOS_GEMM_SPLIT_ACC=1: correction terms share the main accumulator (64 registers
  // less per wave; error = the fp32-MFMA kernel's instead of a third of it)
  static const int nacc = [] {
    const char* e = getenv("EPOS_GEMM_SPLIT_ACC");
    return e ? atoi(e) : 2;
  }();
  if (nacc == 1) {
    if (res) return single ? launch_split_tt<true, true, CB, false>(g, total, s)
                           : launch_split_tt<true, false, CB, false>(g, total, s);
    return single ? launch_split_tt<false, true, CB, false>(g, total, s)
                  : launch_split_tt<false, false, CB, false>(g, total, s);
  }
  if (res) return single ? launch_split_tt<true, true, CB, true>(g, total, s)
                         : launch_split_tt<true, false, CB, true>(g, total, s);
  return single ? launch_split_tt<false, true, CB, true>(g, total, s)
                : launch_split_tt<false, false, CB, true>(g, total, s);
}

}  // namespace

// A group goes to the split kernel when every problem carries split weights and has no
// pre-activation ReLU. The choice depends on nothing else (not on N, not on the other
// problems of the group): a layer gives bit-identical results whether it is launched
// alone or grouped (the sparse-head path relies on that).
bool split_eligible(const EposPointwiseArgs* args, int count) {
  static const int mode = [] {
    const char* e = getenv("EPOS_GEMM_SPLIT");
    return e ? atoi(e) : 1;
  }();
  if (mode == 0) return false;
  for (int i = 0; i < count; ++i) {
    const EposPointwiseArgs& a = args[i];
    if (!a.Ws || a.relu_in != 0 || a.M <= 8 || (a.K & 3) != 0 || (a.lda & 3) != 0 ||
        (reinterpret_cast<uintptr_t>(a.A) & 15) != 0)
      return false;
    // the A pieces are addressed as p.A + 32-bit byte offset
    const int64_t rows = a.sub > 1 ? static_cast<int64_t>(a.M) / (static_cast<int64_t>(a.Ho) * a.Wo) *
                                         a.Hi * a.Wi
                                   : a.M;
    if (rows * a.lda * 4 >= (1LL << 32)) return false;
  }
  return true;
}

int launch_grouped_split(const EposPointwiseArgs* args, int count, hipStream_t s,
                         const int* conv_cin, const int* conv_rate) {
  // 128-row tiles (half the W traffic per MFMA, every A value split once) unless the
  // grid would leave a fifth of the CUs without any workgroup; 64-row tiles then. With
  // several steps in flight the other streams' workgroups fill the second slot of a CU,
  // so the larger tile wins from ~200 tiles on (measured end to end: 303 -> 327
  // images/s against the earlier ">= 512 tiles" rule). EPOS_GEMM_SPLIT_ROWS=64|128
  // forces one of them (tuning).
  static const int forced = [] {
    const char* e = getenv("EPOS_GEMM_SPLIT_ROWS");
    return e ? atoi(e) : 0;
  }();
  int64_t tiles128 = 0;
  for (int i = 0; i < count; ++i)
    tiles128 += ceil_div(args[i].M, 128) * ceil_div(args[i].N, SP_BN);
  const bool big = forced ? forced == 128 : tiles128 >= 200;
  if (conv_cin && (count != 1 || args[0].R != nullptr)) {
    set_error("launch_grouped_split: implicit conv = one problem without residual");
    return EPOS_E_INVALID;
  }
  return big ? launch_split_rb<4>(args, count, s, conv_cin, conv_rate)
             : launch_split_rb<2>(args, count, s, conv_cin, conv_rate);
}

}  // namespace epos

extern "C" int64_t epos_pack_pointwise_weights_split(const float* w_kn, int K, int N,
                                                     void* dst) {
  using namespace epos;
  const int64_t tiles_n = ceil_div(N, SP_BN), nks = ceil_div(K, SP_BK);
  const int64_t total = tiles_n * nks * SP_W_BYTES;
  if (!dst) return total;
  uint16_t* out = static_cast<uint16_t*>(dst);
  for (int64_t tn = 0; tn < tiles_n; ++tn)
    for (int64_t ks = 0; ks < nks; ++ks)
      for (int cbw = 0; cbw < 4; ++cbw)
        for (int ln = 0; ln < 64; ++ln)
          for (int j = 0; j < 8; ++j) {
            const int64_t col = tn * SP_BN + cbw * 32 + (ln & 31);
            const int64_t k = ks * SP_BK + (ln >> 5) * 8 + j;
            const float w = (k < K && col < N) ? w_kn[k * static_cast<int64_t>(N) + col] : 0.f;
            uint32_t hb, mb, lb;
            memcpy(&hb, &w, 4);
            float hf; const uint32_t hm = hb & 0xffff0000u; memcpy(&hf, &hm, 4);
            const float r1 = w - hf;
            memcpy(&mb, &r1, 4);
            float mf; const uint32_t mm = mb & 0xffff0000u; memcpy(&mf, &mm, 4);
            const float r2 = r1 - mf;
            memcpy(&lb, &r2, 4);
            const uint16_t piece[3] = {static_cast<uint16_t>(hb >> 16),
                                       static_cast<uint16_t>(mb >> 16),
                                       static_cast<uint16_t>(lb >> 16)};
            for (int pc = 0; pc < 3; ++pc)
              out[((((tn * nks + ks) * 4 + cbw) * 3 + pc) * 64 + ln) * 8 + j] = piece[pc];
          }
  return total;
}
