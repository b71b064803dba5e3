// fp32 pointwise GEMM on the fp16 matrix pipe of gfx950 ("h2" kernel: two fp16 pieces per
// operand, three piece products per fp32 product). Round 3; the default fp32 GEMM.
//
// The bf16 x 6 split kernel (pointwise_gemm_split.hip) spends six 32x32x16 MFMAs per
// fp32-equivalent 32x32x16 block, and the step it dominates runs AT the chip's power cap:
// matrix-pipe work is what the energy goes to. fp16 has 11 significand bits where bf16 has
// 8, so TWO round-to-nearest pieces carry an fp32 significand,
//     t  = x * 2^e                      (power of two: exact)
//     hi = rn_fp16(t)                   |t - hi| <= 2^-11 |t|, t - hi exact in fp32
//     mid = rn_fp16((t - hi) * 2^11)    hi + mid * 2^-11 = t up to 1 ulp of the 24-bit t
// and the product needs THREE MFMAs (v_mfma_f32_32x32x16_f16, same rate as bf16):
//     acc  += ah * bh
//     corr += ah * bm + am * bh         (scaled by 2^-11 when added to acc in the epilogue)
// am * bm (<= 2^-22 |a b|, random sign) is dropped. Piece products are exact in fp32
// (11 x 11 bits), each MFMA adds 16 of them with one rounding; measured against fp64 the
// rms error is ~1.2e-8 of sum |a||w| (the fp32-MFMA kernel: 2.8e-8, the bf16 x 6 kernel:
// 1.1e-8) -- tests/test_gpu_layers.py holds it to the same bars as the split kernel.
//
// What fp16 costs is exponent range (5 bits), so both operands are scaled by powers of two:
//   * W per output column on the host (epos_pack_pointwise_weights_h2: column maximum into
//     [2^14, 2^15); weights down to 2^-28 of their column's maximum keep full precision,
//     smaller ones degrade gracefully -- absolute error <= 2^-50 x column maximum -- exactly
//     like the activation side below. Only Inf / NaN weights and columns whose scale leaves
//     the fp32 exponent range are refused; such a layer keeps the bf16 x 6 kernel);
//   * A per tensor, at run time, from an upper bound of max|A| that the producers of A
//     maintain in device memory (EposPointwiseArgs.a_amax: "absmax slots", atomic max in the
//     GEMM epilogues; for a depthwise output the bound follows from the depthwise input's
//     slot and the filter's l1 norm). The scale puts the BOUND into [2^14, 2^15): no element
//     can overflow, elements down to ~2^-27 of the bound keep full precision, smaller ones
//     degrade gracefully (absolute error <= 2^-50 x bound). The rounding of hi / mid does not
//     depend on the power of two, so results do not depend on how tight the bound is.
// The epilogue multiplies by the two inverse scales (exact) before bias / residual / ReLU.
//
// Structure = the split kernel's: 128 x 128 tile per 256-thread workgroup, waves 4 x 1 (a
// wave owns 32 rows and all 128 columns: 4 column blocks x {acc, corr}), K step 16 per
// stage, LDS-DMA ring of FIVE 16 KB stages (A: 128 rows x 64 B fp32, XOR-swizzled; W: 8 KB
// of lane-linear fp16 fragments) = 80 KB, two workgroups per CU; tile kt+4 is issued while
// tile kt is computed (its four pieces between the first MFMAs), counted vmcnt, one raw
// s_barrier per stage; the next stage's A fragment is split between the MFMAs of the second
// half (7 VALU per pair of values: pk_mul, cvt_pk, 2 cvt, pk_add, pk_mul, cvt_pk). Implicit
// 3x3 conv mode, grouped launches, strided-row shortcut form, XCD-aware / banded tile order
// and the float4 epilogue are the split kernel's.
//
// Round 4: the activation may arrive already split by its producer (PRESPLIT: no conversion in
// the loop); the epilogue publishes the output's absmax with ONE atomic per workgroup (912
// same-line atomics kept a middle-flow launch alive 2 us after its last store), can write
// streaming stores and 32-row block sums. Launch shapes (template parameter NB; same bits in
// both): 128 x 128 tiles, and 128 x 64 tiles for launches of few tiles (up to 100).
//
// Round 5 (diet): what was built on this kernel, measured slower or no better, and REMOVED
// from the product again -- the fused separable conv's producer phase, the fragment softmax
// in the epilogue, eight waves per 128 x 128 tile, the 256 x 128 eight-wave tile, deeper
// rings, other LDS-DMA placements -- is described with its measurements in DESIGN.md (e) and
// profiles/r04, profiles/r05; the code is in the history (last full version: commit 1b05025).
// Also tried in round 5 and removed: loader / consumer wave specialisation (eight waves, four
// of them issuing only the LDS-DMA pieces, four only MFMAs + fragment reads; bit-identical;
// commit 374286e): SLOWER -- 344 vs 384 TFLOP/s in steady state -- one MFMA wave per SIMD
// cannot cover the per-K-step barrier and fragment-read latency by itself, DMA issue or not
// (profiles/r05/gemm_h2_loader_consumer_waves.txt). Likewise eight register-lean waves per
// tile (fragments re-read in place, 128 VGPRs, two workgroups = FOUR MFMA waves per SIMD;
// commit 96bd7f9: pipe utilisation 61.7 vs 67.2 %, end to end -3.5 %) and one barrier per TWO
// K steps (commit 417060e: +-0). All of these are bit-identical and all deliver the same
// ~380 TFLOP/s in steady state although their pipe utilisation in CYCLES differs (62-70 %):
// the socket sits at its 1400 W cap there and the clock gives back what the schedule gains.
// Round 6: (a) the scale of A is requested before and finished behind the prologue's LDS-DMA
// issue in EVERY variant (the residual variants waited for it first); (b) built, bit-identical
// (482 tests), measured and removed: workgroups that walk several tiles (grid = two per CU,
// tile raw, raw + grid, ...; raw barriers without a fence, so that the stores of tile i are
// still in flight while the pieces of tile i + 1 are issued) -- per launch 73.8 -> 74.6 us
// (19200 x 728 x 728), 159.3 -> 161.7 (19200 x 4032 x 256, the heads' shape), 22.8 -> 23.3
// (76800 x 128 x 128); end to end C2 +0.6 %, C3's shard -0.5 %, C5 -0.3 %; and the loop form cost
// the group instantiations 26 VGPRs (224 -> 250-256) and the SINGLE ones, with their pinned
// arguments carried around the loop, 380-816 bytes of scratch. The hardware already overlaps a
// finishing workgroup's store drain with its CU's other workgroup. profiles/r06/ab_persist_*.txt,
// persist_micro.txt; the code is commit 3f47a57.
#include <string.h>

#include <mutex>
#include <utility>

#include "h2_scale.h"
#include "pointwise_gemm.h"

namespace epos {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float h2_f32x4 __attribute__((ext_vector_type(4)));

#ifdef EPOS_GEMM_TRACE      // tools/gemm_h2_trace.py: 100 MHz stamps per workgroup
__device__ uint64_t* g_h2_trace = nullptr;
#define H2_STAMP(i) do { if (g_h2_trace && t == 0) g_h2_trace[8 * static_cast<uint64_t>(blockIdx.x) + (i)] = wall_clock64(); } while (0)
#else
#define H2_STAMP(i) do { } while (0)
#endif
constexpr int RING_DEVICES = 16;
constexpr int H2_BM = 128, H2_BN = 128, H2_BK = 16;
constexpr int H2_W_BYTES = H2_BK * H2_BN * 4;       // 8192: 4 col blocks x 2 pieces x 1 KB
constexpr int H2_A_BYTES = H2_BM * H2_BK * 4;       // 8192
constexpr int H2_STAGE = H2_W_BYTES + H2_A_BYTES;   // 16384
constexpr int H2_NST = 5;
constexpr int H2_LDS = H2_NST * H2_STAGE;           // 81920: two workgroups per CU
constexpr int H2_NP = 4;                            // LDS-DMA pieces per wave and stage
constexpr int H2_EP_ROW = 132;                      // floats per staged epilogue row
// Tile geometry by column blocks per wave: NB = 4 is the 128 x 128 tile above; NB = 2 a
// 128 x 64 tile (round 4: launches of few tiles -- ASPP 1x1, concat projection: ~80 wide
// tiles for 256 CUs -- get twice the workgroups). A 64-column tile reads one half of the packed
// 8 KB W stage image of its 128-column tile; the A stage is the same. (Other launch shapes --
// eight waves on a 128 x 128 tile, a 256 x 128 tile with eight waves, deeper rings -- were
// built, bit-identical, and measured no better: DESIGN.md (e), profiles/r04, profiles/r05.)
template <int NB> struct H2Geo {
  static constexpr int BN = NB * 32;
  static constexpr int W_LDS = BN * 64;                    // W bytes per stage in LDS
  static constexpr int STAGE = W_LDS + H2_A_BYTES;
  static constexpr int LDS = H2_NST * STAGE;               // 81920 / 61440 (five stages)
  static constexpr int NA = 2;                             // A pieces per wave and stage
  static constexpr int NWP = NB / 2;                       // W pieces per wave and stage
  static constexpr int NP = NA + NWP;                      // LDS-DMA pieces per wave and stage
  static constexpr int EP_ROW = NB * 32 + 4;               // a wave stages its own columns
};
static_assert(H2Geo<4>::LDS == H2_LDS && H2Geo<4>::NP == H2_NP && H2Geo<4>::EP_ROW == H2_EP_ROW, "");
template <int... I, class F>
__device__ __forceinline__ void h2_static_for(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N> __device__ __forceinline__ void h2_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void h2_wait_vm_lgkm0() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(N) : "memory");
}

__device__ __forceinline__ void mfma_f16(const u32x4& a, const u32x4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                             __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void split_pair(float x0, float x1, float s, unsigned& hi,
                                           unsigned& mid) {
  h2_split_pair(x0, x1, s, hi, mid);
}
__device__ __forceinline__ void a_scale(const EposPointwiseArgs& p, int lane, float& s,
                                        float& inv) {
  h2_scale(p.a_amax, p.a_amax2, p.a_gain, p.a_bias, lane, s, inv);
}

struct H2Div { unsigned mul, sh1, sh2; };          // n / d by multiply-shift (32-bit n)
__device__ __forceinline__ unsigned h2_div(unsigned n, const H2Div& f) {
  const unsigned t = __umulhi(f.mul, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// Epilogue of the h2 kernel: value = (acc + corr * 2^-11) * 2^-e_n * 2^-e_a + bias
// (+ residual) (ReLU), transposed through the wave's LDS region and written as float4
// rows (see vec_epilogue in pointwise_gemm.h); optionally max|value| -> c_amax.
template <bool HAS_RES, int NB>
__device__ __forceinline__ float vec_epilogue_h2(float* ws, const f32x16* acc,
                                                const f32x16* corr, const float* cn,
                                                const float* bias4, float inv_a,
                                                 const EposPointwiseArgs& p, int m0w, int n0w,
                                                 int lane) {
  const int l31 = lane & 31, h = lane >> 5;
  const int M = p.M, N = p.N;
  constexpr int EPR = H2Geo<NB>::EP_ROW;
  constexpr int C4 = NB * 8;                 // float4 per staged row (128 / 64 columns)
  constexpr int RPI = 64 / C4;               // rows per wave instruction
  constexpr int NI = 32 / RPI;
  const int c4 = lane & (C4 - 1);
  int r0 = lane / C4;
  // opaque, and behind the K loop's own asm statements: otherwise the sixteen 64-bit row
  // addresses of the residual / the output (loop invariant) are computed BEFORE the K loop
  // and live through it -- the residual variants sat at 256 VGPRs with a spill because of it
  asm volatile("" : "+v"(r0));
  const int n = n0w + c4 * 4;
  // The residual rows are requested in TWO halves: the first before anything is staged, the
  // second once the first half of the column blocks has been staged -- their accumulators
  // are dead by then, so the second half's 32 registers do not come on top of all 128
  // accumulator registers (requested all at once, the residual + pre-split variant sat at
  // 256 VGPRs with a spill; round 5).
  float4 rv[HAS_RES ? NI : 1];
  const int ncl = n < N ? n : 0;
  auto request_res = [&](int i_lo, int i_hi) {
#pragma unroll
    for (int i = i_lo; i < i_hi; ++i) {
      int m = m0w + r0 + i * RPI;
      m = m < M ? m : M - 1;
      rv[i] = *reinterpret_cast<const float4*>(p.R + static_cast<int64_t>(m) * p.ldr + ncl);
    }
  };
  if (HAS_RES) request_res(0, NI / 2);
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (HAS_RES && j == NB / 2) request_res(NI / 2, NI);
    // residual variants (bias4 == nullptr): the bias is added in the row phase below -- a
    // bias load issued here sits behind the residual rows in the in-order memory queue and
    // made this step wait for all of them (3.1 instead of 1.0 us in the trace)
    const float bias = bias4 ? bias4[j] : 0.f;
    const float c = cn[j];
    // two rows at a time (v_pk_fma / v_pk_mul / v_pk_add: the same operations per element)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const h2_f32x2 a2 = {acc[j][r], acc[j][r + 1]}, c2 = {corr[j][r], corr[j][r + 1]};
      const h2_f32x2 k2 = {0x1p-11f, 0x1p-11f}, cc = {c, c}, ia = {inv_a, inv_a}, b2 = {bias, bias};
      const h2_f32x2 sum = __builtin_elementwise_fma(c2, k2, a2);
      h2_f32x2 sv = (sum * cc) * ia;
      if (bias4) sv = sv + b2;
      ws[((r & 3) + 8 * (r >> 2) + 4 * h) * EPR + j * 32 + l31] = sv[0];
      ws[(((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * h) * EPR + j * 32 + l31] = sv[1];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#ifdef EPOS_GEMM_TRACE
  if (g_h2_trace && threadIdx.x == 0) g_h2_trace[8 * static_cast<uint64_t>(blockIdx.x) + 6] = wall_clock64();
#endif
  const bool relu = p.relu != 0;
  float amax = 0.f;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!bias4 && p.bias) {                    // same value, same order: (x + bias) + residual
    const float* bsrc = p.bias + (n < N ? n : 0);     // padded to a multiple of 128 floats
    bv = *reinterpret_cast<const float4*>(bsrc);
  }
  // Row phase. All sixteen staged rows are read first (the accumulators are dead: 64 VGPRs to
  // spare), the ReLU test is hoisted out of the loop and the row address advances by one
  // 64-bit add: the first version waited for every ds_read on the spot, tested `relu` and
  // multiplied m * ldc per row -- 2.0 us of every launch (profiles/r04).
  float4 v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
    v[i] = *reinterpret_cast<const float4*>(ws + (r0 + i * RPI) * EPR + c4 * 4);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (!bias4) { v[i].x += bv.x; v[i].y += bv.y; v[i].z += bv.z; v[i].w += bv.w; }
    if (HAS_RES) {
      v[i].x += rv[i].x; v[i].y += rv[i].y; v[i].z += rv[i].z; v[i].w += rv[i].w;
    }
  }
  if (relu) {
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = relu4(v[i]);
  }
  if (!HAS_RES && p.col_sums) {
    // column sums of this wave's 32 stored rows (image pooling, model.py:220): rows in the
    // order the lane holds them, then the two half-waves (even / odd rows); lanes 0..31 write
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (m0w + r0 + i * RPI < M) { cs.x += v[i].x; cs.y += v[i].y; cs.z += v[i].z; cs.w += v[i].w; }
    }
    if constexpr (NB == 2) {
      cs.x += __shfl_xor(cs.x, 16, 64); cs.y += __shfl_xor(cs.y, 16, 64);
      cs.z += __shfl_xor(cs.z, 16, 64); cs.w += __shfl_xor(cs.w, 16, 64);
    }
    cs.x += __shfl_xor(cs.x, 32, 64); cs.y += __shfl_xor(cs.y, 32, 64);
    cs.z += __shfl_xor(cs.z, 32, 64); cs.w += __shfl_xor(cs.w, 32, 64);
    if (lane < C4 && n < N && m0w < M)
      *reinterpret_cast<float4*>(p.col_sums + static_cast<int64_t>(m0w >> 5) * p.col_ld + n) = cs;
  }
  float* crow = p.C + (static_cast<int64_t>(m0w + r0) * p.ldc + n);
  const int64_t cstep = static_cast<int64_t>(RPI) * p.ldc;
  const bool nok = n < N;
  const bool nt = p.c_stream != 0;              // streaming stores (the dense heads)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = m0w + r0 + i * RPI;
    if (m < M && nok) {
      if (nt) {
        const h2_f32x4 nv = {v[i].x, v[i].y, v[i].z, v[i].w};
        __builtin_nontemporal_store(nv, reinterpret_cast<h2_f32x4*>(crow));
      }
      else *reinterpret_cast<float4*>(crow) = v[i];
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[i].x), fabsf(v[i].y))),
                   fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
    }
    crow += cstep;
  }
#ifdef EPOS_GEMM_TRACE
  if (g_h2_trace && threadIdx.x == 0) g_h2_trace[8 * static_cast<uint64_t>(blockIdx.x) + 7] = wall_clock64();
#endif
  return amax;              // max |stored value| of this lane (the caller publishes it)
}

// PRESPLIT: every problem of the launch has its A operand already as fp16 pairs
// (EposPointwiseArgs.a_presplit; the plan does not mix the two kinds in one group).
template <bool HAS_RES, bool SINGLE, bool CONV, bool PRESPLIT, int NB = 4>
__global__ __launch_bounds__(256, 2)
void pointwise_gemm_h2_f32(GroupedArgs ga_) {
  static_assert(NB == 4 || NB == 2, "tile = 128 x 128 or 128 x 64");
  using Geo = H2Geo<NB>;
  constexpr int NW = 4;                        // waves 4 x 1: a wave owns 32 rows
  constexpr int NST = H2_NST;
  constexpr int BM = H2_BM;
  constexpr int NP = Geo::NP;
  constexpr int NA = Geo::NA;
  constexpr int LA = NST - 1;                  // tiles issued ahead of the one computed
  static_assert((LA - 1) * NP <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wrow = wave;                       // the wave's 32-row group
  const int l31 = lane & 31, h = lane >> 5;

  (void)ga_;
  H2_STAMP(0);
  const GroupedArgs* __restrict__ gp =
      (const GroupedArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  if constexpr (SINGLE) {
    // One round trip for the kernel arguments: left to itself hipcc fetches them where they
    // are first used -- six dependent scalar-load rounds (~0.3 us each, cold scalar cache)
    // before the first LDS-DMA piece could be issued. Pinning the values here puts all the
    // s_loads into this block, behind one wait.
    const EposPointwiseArgs& q = gp->p[0];
    uint64_t a0 = reinterpret_cast<uint64_t>(q.A), a1 = reinterpret_cast<uint64_t>(q.Wh),
             a2 = reinterpret_cast<uint64_t>(q.C), a3 = reinterpret_cast<uint64_t>(q.R),
             a4 = reinterpret_cast<uint64_t>(q.bias), a5 = reinterpret_cast<uint64_t>(q.a_amax),
             a6 = reinterpret_cast<uint64_t>(q.a_amax2), a7 = reinterpret_cast<uint64_t>(q.c_amax),
             a8 = reinterpret_cast<uint64_t>(gp->zero_chunk), l0 = static_cast<uint64_t>(q.lda),
             l1 = static_cast<uint64_t>(q.ldc), l2 = static_cast<uint64_t>(q.ldr);
    int i0 = q.M, i1 = q.N, i2 = q.K, i3 = q.relu, i4 = q.sub, i5 = gp->tiles_n[0],
        i6 = gp->tile_start[MAX_GROUP], i7 = q.Ho, i8 = q.Wo, i9 = q.Hi, i10 = q.Wi,
        i11 = q.c_stream;
    uint64_t a9 = reinterpret_cast<uint64_t>(q.col_sums), l3 = static_cast<uint64_t>(q.col_ld);
    unsigned u0 = gp->tn_mul[0], u1 = gp->tn_sh1[0], u2 = gp->tn_sh2[0];
    float f0 = q.a_gain, f1 = q.a_bias;
    asm volatile("" : : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6), "s"(a7),
                 "s"(a8), "s"(l0), "s"(l1), "s"(l2));
    asm volatile("" : : "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(i4), "s"(i5), "s"(i6), "s"(u0),
                 "s"(u1), "s"(u2), "s"(f0), "s"(f1));
    asm volatile("" : : "s"(i7), "s"(i8), "s"(i9), "s"(i10), "s"(i11), "s"(a9), "s"(l3));
  }
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
  const float* zero_chunk = uniform_ptr(gp->zero_chunk);
  const int M = p.M, N = p.N, K = p.K;
  // tile order inside an XCD's range: column fastest; wide problems in bands of 8 column
  // tiles (all row tiles of a band before the next band), as in the split kernel
  int tile_m, tile_n;
  if (tiles_n <= 8) {
    const H2Div dv = {gp->tn_mul[pi], gp->tn_sh1[pi], gp->tn_sh2[pi]};
    tile_m = static_cast<int>(h2_div(static_cast<unsigned>(bid), dv));
    tile_n = bid - tile_m * tiles_n;
  } else {
    const int tiles_m = (M + BM - 1) / BM;
    const int per_band = tiles_m * 8;
    const int band = bid / per_band;
    const int rem = bid - band * per_band;
    const int left = tiles_n - band * 8;
    const int bw = left < 8 ? left : 8;
    tile_m = rem / bw;
    tile_n = band * 8 + (rem - tile_m * bw);
  }
  const int m0 = tile_m * BM, n0 = tile_n * Geo::BN;
  const int tn128 = Geo::BN == 128 ? tiles_n : (tiles_n + 1) >> 1;   // packed W: 128-column images
  const int n0w = n0;                                                   // this wave's first column
  const int nks = (K + H2_BK - 1) / H2_BK;
  const int cblocks = CONV ? gp->conv_cin[pi] / H2_BK : 1;   // channel blocks per tap
  const int crate = CONV ? gp->conv_rate[pi] : 1;

  // The scale of A comes from a global load (the absmax slot) + a wave reduction: ~2 us of
  // latency that the first LDS-DMA stages can hide -- so it is computed AFTER the prologue's
  // DMA issue (below).
  // (Only in the variants with registers to spare: the residual ones sit at 256 VGPRs.)
  constexpr bool LATE_SCALE = true;
  constexpr bool EARLY_EPI = !HAS_RES;
  float sa_v = 0.f, inv_a = 0.f, sa = 0.f;
  unsigned am_raw = 0, am_raw2 = 0;
  if constexpr (LATE_SCALE) {
    h2_scale_load(p.a_amax, p.a_amax2, lane, am_raw, am_raw2);
  } else {
    a_scale(p, lane, sa_v, inv_a);
    sa = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sa_v)));
  }
  // The epilogue's per-column operands (inverse weight scales, bias) are requested here
  // too -- ahead of the LDS-DMA pieces, so that they retire first and their latency does
  // not stand between the K loop and the stores (8 VGPRs).
  float cn[NB], bias4[NB];
  if constexpr (EARLY_EPI) {
    const float* cscale = reinterpret_cast<const float*>(
        static_cast<const char*>(p.Wh) + static_cast<int64_t>(tn128) *
                                             ((K + H2_BK - 1) / H2_BK) * H2_W_BYTES);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int nb = n0w + j * 32 + l31;
      cn[j] = cscale[nb];                                 // padded to tiles_n * 128
      // unconditional load (no bias: a zero word), so that nothing waits for it here
      const float* bsrc = p.bias ? p.bias + (nb < N ? nb : N - 1) : zero_chunk;
      bias4[j] = *bsrc;
    }
  }

  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) float*)smem));

  // ---- A pieces (1 KB = 16 rows x 64 B): piece = wave*2 + i, lane -> (row, slot);
  //      slot s of row r holds chunk s ^ ((r >> 2) & 3)
  const float* asrc[2];
  unsigned avoff[2];
  int achunk[2];
  int apy[CONV ? 2 : 1], apx[CONV ? 2 : 1];
  unsigned a_dst[2];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = 16 * (wave * NA + i) + (lane >> 2);
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
    // 32-bit element offset (h2_eligible: every row read lies within 4 GB of p.A)
    const unsigned eoff = static_cast<unsigned>(row) * static_cast<unsigned>(p.lda) + c * 4;
    asrc[i] = p.A + eoff;
    avoff[i] = eoff * 4u;                                            // bytes from p.A
    achunk[i] = c * 4;
    a_dst[i] = lds0 + Geo::W_LDS + (wave_u * NA + i) * 1024;
  }
  // ---- W pieces: the 8 KB stage image is contiguous in the packed buffer
  //      (NB = 2: one piece per wave out of the tile's half of the image)
  const unsigned wvoff = static_cast<unsigned>(((wave * Geo::NWP) * 64 + lane) * 16);
  const unsigned w_dst = lds0 + (wave_u * Geo::NWP) * 1024;
  const float* abase = uniform_ptr(p.A);
  const float* wsb = uniform_ptr(reinterpret_cast<const float*>(
      static_cast<const char*>(p.Wh) +
      (Geo::BN == 128 ? static_cast<int64_t>(tile_n) * nks * H2_W_BYTES
                      : static_cast<int64_t>(tile_n >> 1) * nks * H2_W_BYTES + (tile_n & 1) * 4096)));

  auto issue_piece = [&](int kt, int stage, auto piece_tag, auto tail_tag) {
    constexpr int PIECE = decltype(piece_tag)::value;
    constexpr bool TAIL = decltype(tail_tag)::value;
    const unsigned so = static_cast<unsigned>(stage) * Geo::STAGE;
    if constexpr (PIECE < 2) {
      const float* src;
      if constexpr (CONV) {
        const int tap = kt / cblocks, cb = kt - tap * cblocks;       // uniform
        const int ky = tap / 3, dy = (ky - 1) * crate, dx = (tap - ky * 3 - 1) * crate;
        const bool ok = static_cast<unsigned>(apy[PIECE] + dy) < static_cast<unsigned>(p.Hi) &&
                        static_cast<unsigned>(apx[PIECE] + dx) < static_cast<unsigned>(p.Wi);
        src = asrc[PIECE] + ((dy * p.Wi + dx) * p.lda + cb * H2_BK);
        src = ok ? src : zero_chunk;
      } else if constexpr (!TAIL) {
        // full K step of a 1x1 conv: scalar base + 32-bit lane offset
        if constexpr (PIECE == 0) {
          glds16_s_m0(avoff[PIECE], abase + kt * H2_BK, a_dst[PIECE] + so);
        } else {
          const float* ab = abase + (kt * H2_BK - PIECE * 256);      // uniform
          glds16_s_off<PIECE * 1024>(avoff[PIECE], ab);
        }
        return;
      } else {
        src = asrc[PIECE] + kt * H2_BK;
        src = (kt * H2_BK + achunk[PIECE] < K) ? src : zero_chunk;
      }
      if constexpr (PIECE == 0) glds16_v_m0(src, a_dst[PIECE] + so);
      else glds16_v_off<PIECE * 1024>(src - PIECE * 256);
    } else {
      const float* wb = wsb + static_cast<int64_t>(kt) * (H2_W_BYTES / 4);
      if constexpr (PIECE == 2) glds16_s_m0(wvoff, wb, w_dst + so);
      else glds16_s_off<1024>(wvoff, wb);        // shares the M0 write of piece 2
    }
  };
  auto issue = [&](int kt, int stage) {
    issue_piece(kt, stage, std::integral_constant<int, 0>{}, std::true_type{});
    if constexpr (NA == 2) issue_piece(kt, stage, std::integral_constant<int, 1>{}, std::true_type{});
    issue_piece(kt, stage, std::integral_constant<int, 2>{}, std::true_type{});
    if constexpr (Geo::NWP == 2) issue_piece(kt, stage, std::integral_constant<int, 3>{}, std::true_type{});
  };

  // ---- fragment addresses (float index from the stage base)
  int a_off[2];
  {
    const int sw = (l31 >> 2) & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      a_off[j] = Geo::W_LDS / 4 + (wrow * 32 + l31) * H2_BK + (((2 * h + j) ^ sw) << 2);
  }
  const int b_off = lane * 4;   // + (cb*2 + piece) * 256 floats

  float4 xa[2];             // raw fp32 A fragments of the NEXT stage to compute
  u32x4 bp[4][2];           // W fragments {hi, mid} per column block (NB of them live)
  auto read_a = [&](int stage) {
    const float* s = smem + stage * (Geo::STAGE / 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) xa[j] = *reinterpret_cast<const float4*>(s + a_off[j]);
  };
  // A already split by its producer (EposPointwiseArgs.a_presplit): a 16-byte chunk holds
  // [4 hi | 4 mid] fp16 of four consecutive k, so the lane's two chunks give the MFMA
  // operands directly -- four 8-byte reads into the halves of (hi, mid), no conversion
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto read_a_ps = [&](int stage, u32x4& hi, u32x4& mid) {
    const float* s = smem + stage * (Geo::STAGE / 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const u32x2 h2v = *reinterpret_cast<const u32x2*>(s + a_off[j]);
      const u32x2 m2v = *reinterpret_cast<const u32x2*>(s + a_off[j] + 2);
      hi[2 * j] = h2v[0]; hi[2 * j + 1] = h2v[1];
      mid[2 * j] = m2v[0]; mid[2 * j + 1] = m2v[1];
    }
  };
  auto read_b = [&](int stage, auto cb_tag) {
    constexpr int cb = decltype(cb_tag)::value;
    const float* s = smem + stage * (Geo::STAGE / 4);
#pragma unroll
    for (int pc = 0; pc < 2; ++pc)
      bp[cb][pc] = *reinterpret_cast<const u32x4*>(s + b_off + (cb * 2 + pc) * 256);
  };

  f32x16 acc[4], corr[4];   // NB of them live
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; corr[j][r] = 0.f; }

  // ---- prologue: up to four tiles in flight, tile 0 landed + visible
  issue(0, 0);
  h2_static_for(std::make_integer_sequence<int, LA - 1>{}, [&](auto i_tag) {
    constexpr int i = decltype(i_tag)::value + 1;
    if (nks > i) issue(i, i);
  });
  H2_STAMP(1);
  if constexpr (LATE_SCALE) {
    h2_scale_finish(am_raw, am_raw2, p.a_gain, p.a_bias, sa_v, inv_a);
    sa = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sa_v)));
  }
  {   // tile 0 has landed once only the later tiles' pieces are outstanding
    bool waited = false;
    h2_static_for(std::make_integer_sequence<int, LA>{}, [&](auto i_tag) {
      constexpr int i = LA - 1 - decltype(i_tag)::value;          // LA-1 .. 0
      if (!waited && nks > i) { h2_wait_vm<i * NP>(); waited = true; }
    });
  }
  __builtin_amdgcn_s_barrier();
  H2_STAMP(2);
  u32x4 ah, am;
  if constexpr (PRESPLIT) {
    read_a_ps(0, ah, am);
  } else {
    read_a(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* x = reinterpret_cast<const float*>(&xa[u >> 1]);
      unsigned hh, mm;
      split_pair(x[(u & 1) * 2], x[(u & 1) * 2 + 1], sa, hh, mm);
      ah[u] = hh; am[u] = mm;
    }
  }
  read_b(0, std::integral_constant<int, 0>{});
  read_b(0, std::integral_constant<int, 1>{});
  if constexpr (NB == 4) {
    read_b(0, std::integral_constant<int, 2>{});
    read_b(0, std::integral_constant<int, 3>{});
  }
  // MODE 0: issue tile kt+LA (full)   1: issue tile kt+LA (the last, maybe partial)
  //      m >= 2: tile kt+LA+1-m is the last (five stages: 2: kt+3, 3: kt+2, 4: kt+1, 5: kt)
  // LIVE: column blocks that hold any column < N (4, or 3 for the last column tile of
  // e.g. N = 728: every wave of the workgroup then skips the same quarter of its MFMAs)
  auto tile = [&](int kt, int stage, auto mode_tag, auto live_tag, auto ps_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr int LIVE = decltype(live_tag)::value;
    constexpr bool PS = decltype(ps_tag)::value;       // A pre-split: no conversion
    constexpr int LAST = LA + 1;                       // the mode of the last tile
    const int s4 = stage + LA >= NST ? stage + LA - NST : stage + LA;
    const int s1 = stage + 1 >= NST ? stage + 1 - NST : stage + 1;
    u32x4 nh, nm;
    auto split_unit = [&](auto u_tag) {
      constexpr int u = decltype(u_tag)::value;
      const float* x = reinterpret_cast<const float*>(&xa[u >> 1]);
      float x0 = x[(u & 1) * 2], x1 = x[(u & 1) * 2 + 1];
      // the conversions are pure: without an anchor instruction selection emits them right
      // behind the ds_read (and its lgkmcnt wait) instead of behind the MFMA they are
      // meant to hide under
      asm volatile("" : "+v"(x0), "+v"(x1));
      unsigned hh, mm;
#ifdef EPOS_H2_ABL_NOSPLIT          // ablation (tools/bench_gemm_h2_abl.py): wrong results
      hh = __float_as_uint(x0) ^ 0x3c003c00u; mm = __float_as_uint(x1) & 0x3bff3bffu;
#else
      split_pair(x0, x1, sa, hh, mm);
#endif
      nh[u] = hh; nm[u] = mm;
    };
    constexpr bool ISSUE = MODE <= 1;
    // one MFMA of the schedule + what is pinned behind it
    //   DMA >= 0: LDS-DMA piece DMA of tile kt+4 after this MFMA
    //   SPL >= 0: split unit SPL of the next stage's A fragment after this MFMA
    auto step = [&](const u32x4& a, const u32x4& b, f32x16& c, auto dma_tag, auto spl_tag) {
      constexpr int DMA = decltype(dma_tag)::value;
      constexpr int SPL = decltype(spl_tag)::value;
      mfma_f16(a, b, c);
#ifdef EPOS_H2_ABL_NODMA
      constexpr bool kIssue = false;
#else
      constexpr bool kIssue = true;
#endif
      if constexpr (kIssue && ISSUE && DMA >= 0 && DMA < H2_NP) {
        __builtin_amdgcn_sched_barrier(0);
        issue_piece(kt + LA, s4, std::integral_constant<int, DMA>{},
                    std::integral_constant<bool, MODE == 1>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!PS && SPL >= 0 && SPL < 4 && MODE != LAST) {
        __builtin_amdgcn_sched_barrier(0);
        split_unit(std::integral_constant<int, SPL>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using N_ = std::integral_constant<int, -1>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // first half: column blocks 0 and 1 interleaved (consecutive MFMAs never on the same
    // accumulator), small terms first per accumulator; the four DMA pieces ride along
#define H2_D(i) std::integral_constant<int, ((i) < H2_NP ? (i) : -1)>{}   // the four pieces behind the first four MFMAs
    step(ah, bp[0][1], corr[0], H2_D(0), N_{});
    step(ah, bp[1][1], corr[1], H2_D(1), N_{});
    step(am, bp[0][0], corr[0], H2_D(2), N_{});
    step(am, bp[1][0], corr[1], H2_D(3), N_{});
    step(ah, bp[0][0], acc[0], H2_D(4), N_{});
    step(ah, bp[1][0], acc[1], H2_D(5), N_{});
    if constexpr (MODE != LAST) {
      // my reads of this stage are complete (fragments are in registers); my pieces of
      // tile kt+1 have landed once at most the later tiles' pieces are outstanding
      // (of tile kt+LA: the four pieces issued above)
#ifndef EPOS_H2_ABL_NOBAR
      if constexpr (MODE <= 1) h2_wait_vm_lgkm0<(LA - 2) * NP + H2_NP>();
      else h2_wait_vm_lgkm0<(LA - MODE) * NP>();
      __builtin_amdgcn_s_barrier();
#endif
#ifndef EPOS_H2_ABL_NOREAD
      if constexpr (PS) read_a_ps(s1, nh, nm); else read_a(s1);
      read_b(s1, std::integral_constant<int, 0>{});
      read_b(s1, std::integral_constant<int, 1>{});
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    // second half: blocks 2 (and 3); the next stage's A fragment is split behind the MFMAs
    if constexpr (LIVE == 4) {
      step(ah, bp[2][1], corr[2], H2_D(6), N_{});
      step(ah, bp[3][1], corr[3], H2_D(7), I0{});
      step(am, bp[2][0], corr[2], H2_D(8), I1{});
      step(am, bp[3][0], corr[3], N_{}, I2{});
      step(ah, bp[2][0], acc[2], N_{}, I3{});
      step(ah, bp[3][0], acc[3], N_{}, N_{});
    } else {
      step(ah, bp[2][1], corr[2], H2_D(6), I0{});
      step(am, bp[2][0], corr[2], H2_D(7), I1{});
      step(ah, bp[2][0], acc[2], H2_D(8), I2{});
      if constexpr (!PS && MODE != LAST) split_unit(I3{});
    }
#undef H2_D
    if constexpr (MODE != LAST) {
#ifndef EPOS_H2_ABL_NOREAD
      read_b(s1, std::integral_constant<int, 2>{});
      if constexpr (LIVE == 4) read_b(s1, std::integral_constant<int, 3>{});
#endif
      ah = nh; am = nm;
    }
  };
  auto k_loop = [&](auto live_tag, auto ps_tag) {
    using LV = decltype(live_tag);
    using PS = decltype(ps_tag);
    using M0 = std::integral_constant<int, 0>;
    int kt = 0;
    for (; kt + 2 * NST - 1 < nks; kt += NST) {        // every LDS offset an immediate
      h2_static_for(std::make_integer_sequence<int, NST>{}, [&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        tile(kt + i, i, M0{}, LV{}, PS{});
      });
    }
    int stage = 0;                          // kt is a multiple of NST here
    auto next = [&] { stage = stage + 1 == NST ? 0 : stage + 1; ++kt; };
    for (; kt + LA + 1 < nks;) { tile(kt, stage, M0{}, LV{}, PS{}); next(); }
    h2_static_for(std::make_integer_sequence<int, LA>{}, [&](auto i_tag) {
      constexpr int m = decltype(i_tag)::value + 1;                  // modes 1 .. LA
      if (kt + LA + 2 - m == nks) { tile(kt, stage, std::integral_constant<int, m>{}, LV{}, PS{}); next(); }
    });
    tile(kt, stage, std::integral_constant<int, LA + 1>{}, LV{}, PS{});
  };
  // ---- NB = 2 (128 x 64 tile): six MFMAs per stage and wave. There is no second half to
  // read the next stage's fragments under, so the order is turned round: wait + barrier at
  // the TOP of tile kt (my pieces of tile kt+1 have landed once at most tiles kt+2, kt+3 are
  // outstanding), the fragments of tile kt+1 are requested into a second register set, then
  // the six MFMAs of tile kt run with the three pieces of tile kt+4 and the split of the
  // next A fragment behind them. Tile kt+4 lands in the stage of tile kt-1, whose reads every
  // wave completed before the barrier of tile kt-1.
  auto tile2 = [&](int kt, int stage, auto mode_tag, auto ps_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    constexpr bool PS = decltype(ps_tag)::value;
    constexpr int LAST = LA + 1;
    const int s4 = stage + LA >= NST ? stage + LA - NST : stage + LA;
    const int s1 = stage + 1 >= NST ? stage + 1 - NST : stage + 1;
    u32x4 nh, nm, nb[2][2];
    if constexpr (MODE != LAST) {
      if constexpr (MODE <= 2) h2_wait_vm_lgkm0<(LA - 2) * NP>();
      else h2_wait_vm_lgkm0<(LA - MODE) * NP>();
      __builtin_amdgcn_s_barrier();
      const float* sb = smem + s1 * (Geo::STAGE / 4);
      if constexpr (PS) read_a_ps(s1, nh, nm); else read_a(s1);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
          nb[cb][pc] = *reinterpret_cast<const u32x4*>(sb + b_off + (cb * 2 + pc) * 256);
      __builtin_amdgcn_sched_barrier(0);
    }
    auto split_unit = [&](auto u_tag) {
      constexpr int u = decltype(u_tag)::value;
      const float* x = reinterpret_cast<const float*>(&xa[u >> 1]);
      float x0 = x[(u & 1) * 2], x1 = x[(u & 1) * 2 + 1];
      asm volatile("" : "+v"(x0), "+v"(x1));         // anchored behind its MFMA (see tile)
      unsigned hh, mm;
      split_pair(x0, x1, sa, hh, mm);
      nh[u] = hh; nm[u] = mm;
    };
    auto step = [&](const u32x4& a, const u32x4& b, f32x16& c, auto dma_tag, auto spl_tag) {
      constexpr int DMA = decltype(dma_tag)::value;
      constexpr int SPL = decltype(spl_tag)::value;
      mfma_f16(a, b, c);
      if constexpr (MODE <= 1 && DMA >= 0 && DMA < NP) {
        __builtin_amdgcn_sched_barrier(0);
        issue_piece(kt + LA, s4, std::integral_constant<int, (DMA < NA ? DMA : 2 + DMA - NA)>{},
                    std::integral_constant<bool, MODE == 1>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (!PS && SPL >= 0 && SPL < 4 && MODE != LAST) {
        __builtin_amdgcn_sched_barrier(0);
        split_unit(std::integral_constant<int, SPL>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    using N_ = std::integral_constant<int, -1>;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    step(ah, bp[0][1], corr[0], I0{}, N_{});
    step(ah, bp[1][1], corr[1], I1{}, N_{});
    step(am, bp[0][0], corr[0], I2{}, I0{});
    step(am, bp[1][0], corr[1], N_{}, I1{});
    step(ah, bp[0][0], acc[0], N_{}, I2{});
    step(ah, bp[1][0], acc[1], N_{}, I3{});
    if constexpr (MODE != LAST) {
      ah = nh; am = nm;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) { bp[cb][0] = nb[cb][0]; bp[cb][1] = nb[cb][1]; }
    }
  };
  auto k_loop2 = [&](auto ps_tag) {
    using PS = decltype(ps_tag);
    using M0 = std::integral_constant<int, 0>;
    int kt = 0;
    for (; kt + 2 * NST - 1 < nks; kt += NST) {        // every LDS offset an immediate
      h2_static_for(std::make_integer_sequence<int, NST>{}, [&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        tile2(kt + i, i, M0{}, PS{});
      });
    }
    int stage = 0;
    auto next = [&] { stage = stage + 1 == NST ? 0 : stage + 1; ++kt; };
    for (; kt + LA + 1 < nks;) { tile2(kt, stage, M0{}, PS{}); next(); }
    h2_static_for(std::make_integer_sequence<int, LA>{}, [&](auto i_tag) {
      constexpr int m = decltype(i_tag)::value + 1;
      if (kt + LA + 2 - m == nks) { tile2(kt, stage, std::integral_constant<int, m>{}, PS{}); next(); }
    });
    tile2(kt, stage, std::integral_constant<int, LA + 1>{}, PS{});
  };
  if constexpr (NB == 4) {
    if (n0 + 96 >= N) k_loop(std::integral_constant<int, 3>{}, std::integral_constant<bool, PRESPLIT>{});
    else k_loop(std::integral_constant<int, 4>{}, std::integral_constant<bool, PRESPLIT>{});
  } else {
    k_loop2(std::integral_constant<bool, PRESPLIT>{});
  }

  H2_STAMP(3);
  // ---- epilogue --------------------------------------------------------------
#ifdef EPOS_H2_ABL_NOEPI            // ablation (tools/power_components_h2.py): no epilogue
  if (p.ldr != 0x7fffffff) return;  // (always taken; the compiler cannot know)
#endif
  if constexpr (!EARLY_EPI) {
    // (requesting these at the top of the last K step was tried: the residual variants sit
    // at 256 VGPRs and spill)
    const float* cscale = reinterpret_cast<const float*>(
        static_cast<const char*>(p.Wh) + static_cast<int64_t>(tn128) * nks * H2_W_BYTES);
#pragma unroll
    for (int j = 0; j < NB; ++j) cn[j] = cscale[n0w + j * 32 + l31]; // padded to tn128*128
  }
  if (vec_epilogue_ok(p, HAS_RES)) {
    __syncthreads();
    H2_STAMP(5);
    float* ws = smem + wave * 32 * Geo::EP_ROW;
    float amax = vec_epilogue_h2<HAS_RES, NB>(ws, acc, corr, cn, EARLY_EPI ? bias4 : nullptr,
                                              inv_a, p, m0 + wrow * 32, n0w, lane);
#ifdef EPOS_H2_AMAX_PER_WAVE      // A/B: the former one-atomic-per-wave publish
    if (p.c_amax) { amax_publish(p.c_amax, amax, lane, blockIdx.x * NW + wave); return; }
#endif
    if (p.c_amax) {
      // ONE atomic per workgroup: the workgroups of a launch finish together, and their
      // atomics all land on the slot's two cache lines -- one per wave (912 for a middle-flow
      // launch) kept the launch alive ~2 us after its last store (profiles/r04).
      // (shuffles through an OPAQUE copy of the lane id: __shfl_xor's six permute indices are
      // otherwise shared with the scale reduction of the prologue and kept in registers
      // through the whole K loop -- the residual variants spilled one of them)
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      auto xor_max = [&](float x, int o) {
        return fmaxf(x, __int_as_float(__builtin_amdgcn_ds_bpermute((lane_e ^ o) << 2,
                                                                    __float_as_int(x))));
      };
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = xor_max(amax, o);
      float* red = smem + NW * 32 * Geo::EP_ROW;           // behind every wave's staged rows
      if (lane == 0) red[wave] = amax;
      __syncthreads();
      if (wave == 0) {
        float v = lane < NW ? red[lane] : 0.f;
#pragma unroll
        for (int o = NW / 2; o > 0; o >>= 1) v = xor_max(v, o);
        if (lane == 0)
          __hip_atomic_fetch_max(p.c_amax + (blockIdx.x & (EPOS_AMAX_WORDS - 1)),
                                 __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#ifdef EPOS_GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stores acknowledged
    H2_STAMP(4);
#endif
    return;
  }
  if constexpr (!EARLY_EPI) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int nb = n0w + j * 32 + l31;
      bias4[j] = p.bias ? p.bias[nb < N ? nb : N - 1] : 0.f;
    }
  }
  const bool relu = p.relu != 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n0w + j * 32 + l31;
    const int nc = n < N ? n : N - 1;
    const float bias = bias4[j];
    const int mb = m0 + wrow * 32 + 4 * h;
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
      float v = fmaf(corr[j][r], 0x1p-11f, acc[j][r]) * cn[j] * inv_a + bias;
      if (HAS_RES) v += rv[r];
      if (relu) v = fmaxf(v, 0.f);
      if (m < M && n < N) p.C[static_cast<int64_t>(m) * p.ldc + n] = v;
    }
  }
}

template <bool HAS_RES, bool SINGLE, bool CONV, bool PRESPLIT = false, int NB = 4>
int launch_h2_tt(const GroupedArgs& g, int total, hipStream_t s) {
  auto kern = pointwise_gemm_h2_f32<HAS_RES, SINGLE, CONV, PRESPLIT, NB>;
  constexpr int lds = H2Geo<NB>::LDS;
  static_assert(lds >= (4 * 32 * H2Geo<NB>::EP_ROW + 4) * 4, "the epilogue stages through the ring");
  static LdsAttrOnce once;
  {
    const int rc = ensure_dynamic_lds(once, reinterpret_cast<const void*>(kern), lds,
                                      "hipFuncSetAttribute(pointwise_gemm_h2_f32)");
    if (rc) return rc;
  }
  // 80 KB (60 KB) per workgroup: at most two per CU = two MFMA waves per SIMD
  hipLaunchKernelGGL(kern, dim3(total), dim3(256), lds, s, g);
  return launch_status("pointwise_gemm_h2_f32");
}

// ---- absmax reduction (epos_absmax_f32 and the library's own measurement of A when the
// caller gave no slot): float4 rows, one atomic per wave.
__global__ __launch_bounds__(256) void absmax_kernel(const float* X, int64_t ldx,
                                                     int64_t rows, int c4n, int cols,
                                                     int vec, unsigned* slot) {
  const int64_t total = rows * c4n;
  float m = 0.f;
  for (int64_t id = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; id < total;
       id += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t r = id / c4n;
    const int c = static_cast<int>(id - r * c4n) * 4;
    const float* px = X + r * ldx + c;
    if (vec && c + 4 <= cols) {
      const float4 v = *reinterpret_cast<const float4*>(px);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    } else {
      for (int e = 0; e < 4 && c + e < cols; ++e) m = fmaxf(m, fabsf(px[e]));
    }
  }
  amax_publish(slot, m, threadIdx.x & 63, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// Zeroing slots is a KERNEL, not hipMemsetAsync: inside a captured hipGraph the memset node
// was observed to run late relative to the kernel nodes that follow it when several graphs
// replay on several streams (ROCm 7.2: the first slot's words were wiped after conv1_1 had
// published into them -- tools/diag_concurrent.py, profiles/r03/), a kernel node is ordered.
__global__ __launch_bounds__(256) void amax_clear_kernel(unsigned* slots, int64_t words) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < words) slots[i] = 0u;
}

int launch_amax_clear(unsigned* slots, int64_t n_slots, hipStream_t s) {
  const int64_t words = n_slots * EPOS_AMAX_WORDS;
  if (words <= 0) return EPOS_OK;
  hipLaunchKernelGGL(amax_clear_kernel, dim3(static_cast<unsigned>(ceil_div(words, 256))),
                     dim3(256), 0, s, slots, words);
  return launch_status("amax_clear_kernel");
}

int launch_absmax(const float* X, int64_t ldx, int64_t rows, int64_t cols, unsigned* slot,
                  hipStream_t s) {
  const int vec = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
  const int c4n = static_cast<int>(ceil_div(cols, 4));
  const int64_t total = rows * c4n;
  if (total <= 0) return EPOS_OK;
  const int blocks = static_cast<int>(total / 256 + 1 < 2048 ? total / 256 + 1 : 2048);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, s, X, ldx, rows, c4n,
                     static_cast<int>(cols), vec, slot);
  return launch_status("absmax_kernel");
}

// The library's own slots for callers that pass Wh without a_amax (single calls, tests):
// a ring of 256 slots, taken round robin; memset + reduction + GEMM are ordered on the
// caller's stream. A slot is reused after 256 further such calls -- plans that overlap
// streams or capture graphs pass their own slots.
// 16 zero bytes per device for GroupedArgs.zero_chunk
// true while `s` is being captured into a graph (an allocation or a legacy-stream memset
// issued then would invalidate the capture)
bool capturing(hipStream_t s) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
}

const float* zero_chunk_dev(hipStream_t s) {
  static std::mutex mu;
  static float* z[RING_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RING_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!z[dev]) {
    // first fp16-pair launch on this device: the chunk is allocated and zeroed NOW -- not
    // possible inside a stream capture (callers warm up once before capturing, as the
    // network plan does); refused with an error instead of breaking the capture
    if (capturing(s)) return nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&z[dev]), 64) != hipSuccess) { z[dev] = nullptr; return nullptr; }
    if (hipMemset(z[dev], 0, 64) != hipSuccess) {     // never hand out an unzeroed chunk
      (void)hipFree(z[dev]);
      z[dev] = nullptr;
      return nullptr;
    }
  }
  return z[dev];
}

// n / tiles_n[i] by multiply-shift for the kernels' tile mapping (Granlund-Montgomery, any
// 32-bit n)
void set_tn_div(GroupedArgs& g, int i) {
  const unsigned d = static_cast<unsigned>(g.tiles_n[i] > 0 ? g.tiles_n[i] : 1);
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  g.tn_mul[i] = static_cast<unsigned>(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  g.tn_sh1[i] = l > 0 ? 1 : 0;
  g.tn_sh2[i] = l > 0 ? l - 1 : 0;
}

constexpr int RING_SLOTS = 256;
unsigned* ring_slot(hipStream_t s) {
  // one ring per device, created under a lock (calls may come from several host threads
  // and devices); the round-robin index is atomic
  static std::mutex mu;
  static unsigned* base[RING_DEVICES] = {};
  static unsigned next[RING_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RING_DEVICES) return nullptr;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!base[dev] && capturing(s)) return nullptr;      // see zero_chunk_dev
    if (!base[dev] &&
        hipMalloc(reinterpret_cast<void**>(&base[dev]),
                  sizeof(unsigned) * EPOS_AMAX_WORDS * RING_SLOTS) != hipSuccess) {
      base[dev] = nullptr;
      return nullptr;
    }
  }
  const unsigned i = __atomic_fetch_add(&next[dev], 1u, __ATOMIC_RELAXED) % RING_SLOTS;
  return base[dev] + static_cast<size_t>(i) * EPOS_AMAX_WORDS;
}

}  // namespace

// A group goes to the h2 kernel when every problem carries fp16-pair weights and has no
// pre-activation ReLU; like split_eligible the choice depends on nothing else, so a layer
// gives the same bits alone or in a group.
bool h2_eligible(const EposPointwiseArgs* args, int count) {
  static const int mode = [] {       // EPOS_GEMM_SPLIT=0 means "fp32 MFMA everywhere"
    const char* e = getenv("EPOS_GEMM_H2");
    const char* sp = getenv("EPOS_GEMM_SPLIT");
    if (sp && atoi(sp) == 0) return 0;
    return e ? atoi(e) : 1;
  }();
  if (mode == 0) return false;
  for (int i = 0; i < count; ++i) {
    const EposPointwiseArgs& a = args[i];
    if (!a.Wh || a.relu_in != 0 || a.M <= 8 || (a.K & 3) != 0 || (a.lda & 3) != 0 ||
        (reinterpret_cast<uintptr_t>(a.A) & 15) != 0)
      return false;
    if (a.a_presplit && !a.a_amax) return false;   // rejected by the entry point
    const int64_t rows = a.sub > 1 ? static_cast<int64_t>(a.M) / (static_cast<int64_t>(a.Ho) * a.Wo) *
                                         a.Hi * a.Wi
                                   : a.M;
    if (rows * a.lda * 4 >= (1LL << 32)) return false;
  }
  return true;
}

int& narrow_limit() {
  static int v = [] {
    const char* e = getenv("EPOS_H2_BN64_MAX_TILES");
    return e ? atoi(e) : 100;
  }();
  return v;
}

int launch_grouped_h2(const EposPointwiseArgs* args, int count, hipStream_t s,
                      const int* conv_cin, const int* conv_rate) {
  if (conv_cin && (count != 1 || args[0].R != nullptr)) {
    set_error("launch_grouped_h2: implicit conv = one problem without residual");
    return EPOS_E_INVALID;
  }
  // one launch = one kind of problem: a group that mixes pre-split and fp32 A operands is
  // refused (the plan never builds one); a group with a residual goes out problem by problem
  // (same bits: an element's value does not depend on the launch it is computed in)
  for (int i = 1; i < count; ++i)
    if ((args[i].a_presplit != 0) != (args[0].a_presplit != 0)) {
      set_error("launch_grouped_h2: the problems of a group must agree on a_presplit");
      return EPOS_E_INVALID;
    }
  if (count > 1)
    for (int i = 0; i < count; ++i)
      if (args[i].R) {
        for (int j = 0; j < count; ++j) {
          const int rc = launch_grouped_h2(args + j, 1, s, nullptr, nullptr);
          if (rc) {       // say which problem failed: the ones before it are already enqueued
            char inner[400];
            snprintf(inner, sizeof(inner), "%s", epos_last_error());
            set_error("epos_pointwise_conv_grouped_f32: problem %d of %d (a group with residuals "
                      "is issued problem by problem; 0..%d are enqueued): %s", j, count, j - 1,
                      inner);
            return rc;
          }
        }
        return EPOS_OK;
      }
  GroupedArgs g = {};
  g.count = count;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    g.p[i] = args[i];
    if (!g.p[i].a_amax) {
      // no bound given: measure A (rows actually read: the whole [B, Hi, Wi] map when
      // the rows are gathered with a stride or by conv taps)
      unsigned* slot = ring_slot(s);
      if (!slot) {
        set_error("launch_grouped_h2: cannot allocate the absmax slot ring (first use on this "
                  "device during a stream capture? launch once before capturing)");
        return EPOS_E_INVALID;
      }
      const bool gathered = conv_cin || args[i].sub > 1;
      const int64_t rows = gathered ? static_cast<int64_t>(args[i].M) /
                                          (static_cast<int64_t>(args[i].Ho) * args[i].Wo) *
                                          args[i].Hi * args[i].Wi
                                    : args[i].M;
      const int64_t cols = conv_cin ? conv_cin[i] : args[i].K;
      int rc = launch_amax_clear(slot, 1, s);
      if (rc) return rc;
      rc = launch_absmax(args[i].A, args[i].lda, rows, cols, slot, s);
      if (rc) return rc;
      g.p[i].a_amax = slot;
      g.p[i].a_amax2 = nullptr;
      g.p[i].a_gain = 0.f;
    }
    g.conv_cin[i] = conv_cin ? conv_cin[i] : 0;
    g.conv_rate[i] = conv_rate ? conv_rate[i] : 1;
  }
  auto lay_out = [&](int bn) {
    total = 0;
    for (int i = 0; i < count; ++i) {
      g.tile_start[i] = total;
      g.tiles_n[i] = static_cast<int>(ceil_div(args[i].N, bn));
      set_tn_div(g, i);
      g.npad[i] = static_cast<int>(ceil_div(args[i].N, H2_BN)) * H2_BN;
      total += static_cast<int>(ceil_div(args[i].M, H2_BM)) * g.tiles_n[i];
    }
    for (int i = count; i <= MAX_GROUP; ++i) g.tile_start[i] = total;
  };
  lay_out(H2_BN);
  // A launch of few 128 x 128 tiles runs as 128 x 64 tiles: twice the workgroups. Measured
  // (profiles/r04/gemm_h2_tile_shapes.txt): worth 20-25 % for launches of ~80 wide tiles
  // (ASPP 1x1, concat projection: two thirds of the CUs idle otherwise), a LOSS from ~150
  // tiles on -- at 228 tiles (a single image's middle-flow layers) two narrow workgroups per
  // CU are slower than one wide one, because a narrow tile moves 1.5x the LDS-DMA bytes per
  // MFMA. Hence the default limit of 100. Same bits either way (an element's K sum does not
  // depend on the tile). EPOS_H2_BN64_MAX_TILES or epos_set_h2_narrow_tile_limit (0 = never).
  bool narrow = !conv_cin && total <= __atomic_load_n(&narrow_limit(), __ATOMIC_RELAXED);
  for (int i = 0; i < count; ++i) narrow = narrow && !args[i].col_sums;
  if (narrow) lay_out(64);
  g.zero_chunk = zero_chunk_dev(s);
  if (!g.zero_chunk) {
    set_error("launch_grouped_h2: cannot allocate the zero chunk (first fp16-pair launch on this "
              "device during a stream capture? launch once before capturing)");
    return EPOS_E_INVALID;
  }
  const bool res = args[0].R != nullptr;
  const bool single = count == 1;
  const bool ps = args[0].a_presplit != 0;
  // Kernel instantiations (11): 128 x 128 tiles -- single problem {residual} x {pre-split A},
  // group {pre-split A}, implicit 3x3 conv; 128 x 64 tiles (always the group form, which also
  // serves one problem) {residual} x {pre-split A}.
  if (conv_cin) return launch_h2_tt<false, true, true>(g, total, s);
  if (narrow) {
    if (ps) return res ? launch_h2_tt<true, false, false, true, 2>(g, total, s)
                       : launch_h2_tt<false, false, false, true, 2>(g, total, s);
    return res ? launch_h2_tt<true, false, false, false, 2>(g, total, s)
               : launch_h2_tt<false, false, false, false, 2>(g, total, s);
  }
  if (single) {
    if (ps) return res ? launch_h2_tt<true, true, false, true>(g, total, s)
                       : launch_h2_tt<false, true, false, true>(g, total, s);
    return res ? launch_h2_tt<true, true, false>(g, total, s)
               : launch_h2_tt<false, true, false>(g, total, s);
  }
  return ps ? launch_h2_tt<false, false, false, true>(g, total, s)
            : launch_h2_tt<false, false, false>(g, total, s);
}

}  // namespace epos

#ifdef EPOS_GEMM_TRACE
extern "C" int epos_debug_set_gemm_trace(uint64_t* buf) {
  return epos::check_hip(hipMemcpyToSymbol(HIP_SYMBOL(epos::g_h2_trace), &buf, sizeof(buf)),
                         "epos_debug_set_gemm_trace");
}
#endif

extern "C" int epos_set_h2_narrow_tile_limit(int max_tiles) {
  return __atomic_exchange_n(&epos::narrow_limit(), max_tiles < 0 ? 0 : max_tiles,
                             __ATOMIC_RELAXED);
}

extern "C" int epos_amax_clear(uint32_t* slots, int64_t n_slots, void* stream) {
  using namespace epos;
  EPOS_REQUIRE(slots && n_slots >= 0, "bad argument");
  return launch_amax_clear(slots, n_slots, static_cast<hipStream_t>(stream));
}

extern "C" int epos_absmax_f32(const float* X, int64_t ldx, int64_t rows, int64_t cols,
                               uint32_t* slot, void* stream) {
  using namespace epos;
  EPOS_REQUIRE(X && slot && rows >= 0 && cols >= 0 && ldx >= cols, "bad argument");
  return launch_absmax(X, ldx, rows, cols, slot, static_cast<hipStream_t>(stream));
}

namespace {
// fp32 -> fp16 round-to-nearest-even (host; the device's v_cvt_pk_f16_f32 in RNE mode)
uint16_t f32_to_f16_rne(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return static_cast<uint16_t>(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (x >= 0x477ff000u) return static_cast<uint16_t>(sign | 0x7c00u);      // >= 65520 -> inf
  if (x < 0x33000001u) return static_cast<uint16_t>(sign);                  // <= 2^-25 -> 0
  const int e = static_cast<int>(x >> 23) - 127;
  uint32_t man = (x & 0x7fffffu) | 0x800000u;                               // 24 bits
  int shift = 13;                                                           // normal: keep 11
  int he = e + 15;
  if (he <= 0) { shift += 1 - he; he = 0; }                                 // denormal
  const uint32_t half = 1u << (shift - 1), rest = man & ((1u << shift) - 1);
  uint32_t q = man >> shift;
  if (rest > half || (rest == half && (q & 1u))) ++q;
  uint32_t out;
  if (he == 0) out = q;                          // may carry into the exponent: correct
  else out = (static_cast<uint32_t>(he - 1) << 10) + q;   // q has the implicit bit at 1<<10
  return static_cast<uint16_t>(sign | out);
}
float f16_to_f32(uint16_t hbits) {
  const uint32_t sign = (hbits & 0x8000u) << 16;
  const int e = (hbits >> 10) & 31;
  const uint32_t m = hbits & 0x3ffu;
  float v;
  if (e == 0) v = ldexpf(static_cast<float>(m), -24);
  else if (e == 31) { const uint32_t b = 0x7f800000u | (m << 13); memcpy(&v, &b, 4); }
  else v = ldexpf(static_cast<float>(m | 0x400u), e - 25);
  uint32_t b;
  memcpy(&b, &v, 4);
  b |= sign;
  memcpy(&v, &b, 4);
  return v;
}
}  // namespace

extern "C" int64_t epos_pack_pointwise_weights_h2(const float* w_kn, int K, int N,
                                                  void* dst) {
  using namespace epos;
  const int64_t tiles_n = ceil_div(N, H2_BN), nks = ceil_div(K, H2_BK);
  const int64_t npad = tiles_n * H2_BN;
  const int64_t wbytes = tiles_n * nks * H2_W_BYTES;
  const int64_t total = wbytes + npad * 4;
  // column scales and the representability check (needs the weights, not dst)
  float* scale = static_cast<float*>(malloc(sizeof(float) * npad));
  bool ok = scale != nullptr;
  for (int64_t n = 0; ok && n < npad; ++n) {
    float cmax = 0.f;
    if (n < N)
      for (int64_t k = 0; k < K; ++k) {
        const float a = fabsf(w_kn[k * static_cast<int64_t>(N) + n]);
        if (!(a <= 3.0e38f)) ok = false;               // Inf / NaN weights: not here
        cmax = a > cmax ? a : cmax;
      }
    int e = 0;
    if (cmax > 0.f) {
      (void)frexpf(cmax, &e);                          // cmax = f * 2^e, f in [0.5, 1)
      e = 15 - e;                                      // cmax * 2^e in [2^14, 2^15)
    }
    if (e > 100 || e < -100) ok = false;
    scale[n] = ldexpf(1.f, e);
  }
  // Every finite weight is representable (round 6; until round 5 a weight more than ~2^27
  // below its column's maximum made the packer refuse the whole matrix): with the column
  // maximum in [2^14, 2^15) a scaled weight t is reproduced to
  //     |hi + mid * 2^-11 - t| <= max(2^-22 |t|, 2^-36)
  // -- full precision down to |t| = 2^-14 (2^-28..2^-29 of the column maximum), and below that
  // hi / mid are fp16 subnormals (the matrix pipe does not flush them) on an absolute grid of
  // 2^-35: an error of at most 2^-50 of the column maximum, which is below the fp32 rounding
  // of any sum the column's larger weights take part in -- the same graceful degradation as
  // on the activation side. The loop below is the packer's self-check of that bound.
  for (int64_t n = 0; ok && n < N; ++n)
    for (int64_t k = 0; k < K; ++k) {
      const float w = w_kn[k * static_cast<int64_t>(N) + n];
      if (w == 0.f) continue;
      const float t = w * scale[n];
      const float hi = f16_to_f32(f32_to_f16_rne(t));
      const float mid = f16_to_f32(f32_to_f16_rne((t - hi) * 2048.f));
      const double err = fabs((static_cast<double>(hi) + static_cast<double>(mid) / 2048.0) - t);
      const double tol = fmax(fabs(static_cast<double>(t)) * 0x1p-22, 0x1p-36);
      if (!(err <= tol)) { ok = false; break; }
    }
  if (!ok) { free(scale); return 0; }
  if (!dst) { free(scale); return total; }
  uint16_t* out = static_cast<uint16_t*>(dst);
  for (int64_t tn = 0; tn < tiles_n; ++tn)
    for (int64_t ks = 0; ks < nks; ++ks)
      for (int cbw = 0; cbw < 4; ++cbw)
        for (int ln = 0; ln < 64; ++ln)
          for (int j = 0; j < 8; ++j) {
            const int64_t col = tn * H2_BN + cbw * 32 + (ln & 31);
            const int64_t k = ks * H2_BK + (ln >> 5) * 8 + j;
            const float w = (k < K && col < N) ? w_kn[k * static_cast<int64_t>(N) + col] : 0.f;
            const float t = w * scale[col];
            const uint16_t hb = f32_to_f16_rne(t);
            const uint16_t mb = f32_to_f16_rne((t - f16_to_f32(hb)) * 2048.f);
            const int64_t base = (((tn * nks + ks) * 4 + cbw) * 2) * 512 + ln * 8 + j;
            out[base] = hb;
            out[base + 512] = mb;
          }
  float* inv = reinterpret_cast<float*>(static_cast<char*>(dst) + wbytes);
  for (int64_t n = 0; n < npad; ++n) inv[n] = 1.f / scale[n];
  free(scale);
  return total;
}
