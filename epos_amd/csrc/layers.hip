// HBM-bound layer kernels of the EPOS network for gfx950 (NHWC fp32):
// depthwise 3x3 (+folded BN, ReLU before/after), im2col for the two dense 3x3
// stem convs, global average pool, bilinear resize (align_corners), grouped
// softmax and argmax. All accesses are float4 along the channel axis (the
// contiguous axis of NHWC), 16 B per lane, so a wave moves 1 KiB per instruction.
#include <stdlib.h>

#include "common.h"
#include "h2_scale.h"

namespace epos {
namespace {

__device__ __forceinline__ float4 ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void st4(float* p, float4 v) {
  *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f),
                     fmaxf(v.w, 0.f));
}
// fp16-pair form of four outputs (EposDepthwiseArgs.y_h2): the same 16 bytes hold
// [hi hi hi hi | mid mid mid mid] of y * s for the fp16-pair GEMM that reads them as its A
typedef unsigned dw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_h2(float* p, float4 v, float s) {
  dw_u32x4 o;
  unsigned h0, h1, m0, m1;
  h2_split_pair(v.x, v.y, s, h0, m0);
  h2_split_pair(v.z, v.w, s, h1, m1);
  o[0] = h0; o[1] = h1; o[2] = m0; o[3] = m1;
  *reinterpret_cast<dw_u32x4*>(p) = o;
}
__device__ __forceinline__ void st4_any(float* p, float4 v, bool h2, float s) {
  if (h2) st4_h2(p, v, s); else st4(p, v);
}
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y),
                     fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// --------------------------------------------------------------------------
// Depthwise 3x3. One thread = one output pixel x 4 channels; consecutive threads
// walk the channel axis first (coalesced), then pixels.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void depthwise3x3_kernel(EposDepthwiseArgs p,
                                                           int c4n,
                                                           int64_t total) {
  EPOS_SET_PRIO(EPOS_DW_PRIO);
  int64_t id = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool h2 = p.y_h2 != 0;                 // uniform
  // the output scale's slot word is requested first and reduced behind the taps' loads (see
  // depthwise3x3_s1_kernel); lanes past the end stay for the wave-wide reduction, recompute
  // the last item and store nothing
  float hs = 1.f;
  unsigned am_raw, am_raw2;
  {
    const unsigned* s1 = h2 ? p.x_amax : reinterpret_cast<const unsigned*>(p.bias);
    const unsigned* s2 = (h2 && p.x_amax2) ? p.x_amax2 : s1;
    const int li = h2 ? static_cast<int>(threadIdx.x & 63) : 0;
    am_raw = s1[li];
    am_raw2 = s2[li];
  }
  const bool live = id < total;
  if (__builtin_amdgcn_ballot_w64(live) == 0) return;
  id = live ? id : total - 1;
  const int c = static_cast<int>(id % c4n) * 4;
  int64_t pix = id / c4n;
  const int xo = static_cast<int>(pix % p.Wo);
  pix /= p.Wo;
  const int yo = static_cast<int>(pix % p.Ho);
  const int b = static_cast<int>(pix / p.Ho);
  const int pad = p.rate;  // SAME (stride 1) and fixed_padding (stride 2) agree
  float4 acc = ld4(p.bias + c);
  const float* xb = p.X + static_cast<int64_t>(b) * p.Hi * p.Wi * p.ldx + c;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yi = yo * p.stride - pad + ky * p.rate;
    if (yi < 0 || yi >= p.Hi) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xi = xo * p.stride - pad + kx * p.rate;
      if (xi < 0 || xi >= p.Wi) continue;
      float4 v = ld4(xb + (static_cast<int64_t>(yi) * p.Wi + xi) * p.ldx);
      if (p.relu_in) v = relu4(v);
      acc = fma4(v, ld4(p.w9c + (ky * 3 + kx) * p.C + c), acc);
    }
  }
  if (p.relu_out) acc = relu4(acc);
  if (h2) {
    float hinv;
    h2_scale_finish(am_raw, am_raw2, p.gain, p.bias0, hs, hinv);
  }
  if (live) st4_any(p.Y + ((static_cast<int64_t>(b) * p.Ho + yo) * p.Wo + xo) * p.ldy + c, acc, h2, hs);
}

// Stride-1 depthwise 3x3 with a SLIDING WINDOW: one thread = 4 channels x a run
// of L output pixels of one row whose x coordinates are `rate` apart (x0, x0+r,
// ...), so consecutive outputs share two of their three input columns. The 9
// weight vectors are loaded once per run and an output costs 3(L+2)/L input
// loads instead of 9 (+9 weights). Loads are unconditional from clamped
// addresses (zero padding is a select) so no load sits under a branch.
//
// XCD-aware partition. Workgroup i runs on XCD i % 8, and each XCD has a private
// 4 MB L2. With a flat id -> (pixel, channel) map every XCD touches the whole
// input (each input row is needed by three output rows that land on different
// XCDs): the tensor streams through all eight L2s from the Infinity Cache, about
// three times in total (PMC: FETCH_SIZE / WRITE_SIZE = 3.4, 12.6 us for the 14 MB
// middle-flow tensor against 4.4 us for a plain copy). A depthwise conv never
// mixes channels, so
//   mode 0 (C >= 256): XCD x owns the channel slice [x*c4n/8, (x+1)*c4n/8) of
//           EVERY pixel -- its share of the input (1/8) stays in its L2 and no
//           byte is fetched by two XCDs;
//   mode 1 (narrow tensors: a slice would be less than a cache line): XCD x owns
//           a band of rows; only the 2*rate halo rows of a band are fetched twice.
//
// The kernel is VALU-issue bound, not bandwidth bound (the first version spent ~900
// vector instructions per wave: 64-bit index arithmetic with runtime divisions, four
// selects per loaded float4 for the zero padding, two v_max per ReLU'd component), so:
// divisions by launch constants use precomputed multipliers, ReLU is one v_med3 per
// component and a compile-time option, and a wave whose lanes all lie in the interior
// of the image (no tap outside) takes a path without any clamp or select.
#ifndef EPOS_DW_REP
#define EPOS_DW_REP 1     // x-runs per thread (they share the nine weight vectors)
#endif
#ifndef EPOS_DW_MIN_BLOCKS
#define EPOS_DW_MIN_BLOCKS 4     // <= 128 VGPRs: see DESIGN.md (co-residency with GEMM waves)
#endif
#ifndef EPOS_DW_MIN_BLOCKS2
#define EPOS_DW_MIN_BLOCKS2 3    // the two-row kernel: <= 168 VGPRs
#endif
struct FastDiv {                 // n / d for any 32-bit n (Granlund-Montgomery)
  unsigned mul, sh1, sh2, d;
};
struct DwPartition {
  int mode;        // 0: channel slices, 1: row bands
  int runs;        // mode 0: runs of the whole launch = B*Ho*nres*nchunk
  int rows;        // mode 1: B*Ho
  int unit;        // mode 0: slices start on multiples of `unit` float4 (8 = one 128-byte line
                   // when rows and base addresses are line-aligned; else 1), nu = units per row
  int nu;
  FastDiv dwc[3];  // mode 0: the (at most three) distinct slice widths
  FastDiv dc4n, dchunk, dres, dho;   // dho: / row slots per image
  FastDiv drate;                     // ROWS = 2: / rate (slot -> row group)
};
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  const unsigned t = __umulhi(f.mul, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
// One v_max per component: fmaxf() costs two under IEEE mode (a canonicalising
// v_max v,v,v first), and hipcc folds fmed3(v, 0, inf) back into the same pair.
__device__ __forceinline__ float relu_1op(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float4 relu4_1op(float4 v) {
  return make_float4(relu_1op(v.x), relu_1op(v.y), relu_1op(v.z), relu_1op(v.w));
}

// ROWS = 2 (round 2): a thread computes the run for TWO output rows `rate` apart (y0 and
// y0 + rate). They share two of their three input rows, so the 4 x (L + 2) loaded float4
// serve 2L outputs: 3.0 input loads per output instead of 4.5, and the nine weight
// vectors are amortised over 8 outputs instead of 4 -- the kernel is bound by load
// instructions issued (one 1 KB wave-load per ~24 ns per CU -- a latency figure at this
// kernel's occupancy: from cache a CU sustains one per 8-12 ns, round 4), not by bytes. Row slots:
// rows are taken in groups of 2 * rate; slot (g, i), i < rate, owns rows 2*rate*g + i and
// + rate. Same fmaf chain per output as ROWS = 1: identical bits.
template <int L, bool RELU_IN, bool RELU_OUT, int ROWS>
__global__ __launch_bounds__(256, (ROWS == 2 ? EPOS_DW_MIN_BLOCKS2 : EPOS_DW_MIN_BLOCKS)) void depthwise3x3_s1_kernel(
    EposDepthwiseArgs p, int c4n, int nres, int nchunk, int nrows, DwPartition part) {
  constexpr int NR = ROWS + 2;                 // input rows held per column
  EPOS_SET_PRIO(EPOS_DW_PRIO);
  {
    // One round trip for the kernel arguments (~70 words): left to itself the compiler
    // fetches them where they are first used -- seven dependent scalar-load rounds in front
    // of the first vector load of a kernel that lives ~10 us (as in the fp16-pair GEMM).
    uint64_t a0 = reinterpret_cast<uint64_t>(p.X), a1 = reinterpret_cast<uint64_t>(p.w9c),
             a2 = reinterpret_cast<uint64_t>(p.bias), a3 = reinterpret_cast<uint64_t>(p.Y),
             a4 = reinterpret_cast<uint64_t>(p.x_amax), a5 = reinterpret_cast<uint64_t>(p.x_amax2),
             l0 = static_cast<uint64_t>(p.ldx), l1 = static_cast<uint64_t>(p.ldy);
    int i0 = p.B, i1 = p.Hi, i2 = p.Wi, i3 = p.Ho, i4 = p.Wo, i5 = p.C, i6 = p.rate, i7 = p.y_h2;
    float f0 = p.gain, f1 = p.bias0;
    const unsigned bd = blockDim.x;               // an implicit argument behind the explicit ones
    asm volatile("" : : "s"(bd));
    asm volatile("" : : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(l0), "s"(l1),
                 "s"(i0), "s"(i1), "s"(i2), "s"(i3), "s"(i4), "s"(i5), "s"(i6), "s"(i7), "s"(f0),
                 "s"(f1), "s"(c4n), "s"(nres), "s"(nchunk), "s"(nrows));
    asm volatile("" : : "s"(part.mode), "s"(part.runs), "s"(part.rows), "s"(part.dwc[0].mul),
                 "s"(part.dwc[0].sh1), "s"(part.dwc[0].sh2), "s"(part.dwc[0].d),
                 "s"(part.dwc[1].mul), "s"(part.dwc[1].sh1), "s"(part.dwc[1].sh2), "s"(part.dwc[1].d),
                 "s"(part.dwc[2].mul), "s"(part.dwc[2].sh1), "s"(part.dwc[2].sh2), "s"(part.unit), "s"(part.nu),
                 "s"(part.dc4n.mul), "s"(part.dc4n.sh1), "s"(part.dc4n.sh2), "s"(part.dchunk.mul),
                 "s"(part.dchunk.sh1), "s"(part.dchunk.sh2), "s"(part.dres.mul), "s"(part.dres.sh1),
                 "s"(part.dres.sh2), "s"(part.dho.mul), "s"(part.dho.sh1), "s"(part.dho.sh2),
                 "s"(part.drate.mul), "s"(part.drate.sh1), "s"(part.drate.sh2));
  }
  const bool h2 = p.y_h2 != 0;                 // uniform: fp16-pair output
  // The scale of an fp16-pair output comes from the absmax slot of X: a global load + a wave
  // reduction. Only the stores need it, so the slot word is REQUESTED here -- unconditionally
  // (a load under a branch is waited for on the spot; without y_h2 the lane reads a word of the
  // bias vector that nobody uses) -- and the reduction runs behind the input loads
  // (dw_scale_finish below). Until round 5 the whole of h2_scale stood here: slot load, wait,
  // six dependent permutes, and only then the weight and input loads -- two dependent memory
  // round trips per launch instead of one (round 6; profiles/r06/dw_scale_round_trip.txt).
  float hs = 1.f;
  unsigned am_raw, am_raw2;
  {
    const unsigned* s1 = h2 ? p.x_amax : reinterpret_cast<const unsigned*>(p.bias);
    const unsigned* s2 = (h2 && p.x_amax2) ? p.x_amax2 : s1;
    const int li = h2 ? static_cast<int>(threadIdx.x & 63) : 0;
    am_raw = s1[li];
    am_raw2 = s2[li];
  }
  static_assert(EPOS_DW_REP == 1, "the scale reduction sits inside the (one) run of a thread");
  auto dw_scale_finish = [&] {
    if (h2) {
      float hinv;
      // opaque: otherwise the first max of the two words is hoisted above the input loads
      // (common to both paths) and takes the wait for the slot with it
      unsigned r1 = am_raw, r2 = am_raw2;
      asm volatile("" : "+v"(r1), "+v"(r2));
      h2_scale_finish(r1, r2, p.gain, p.bias0, hs, hinv);
    }
  };
  const int xcd = blockIdx.x & 7;
  const unsigned local = (blockIdx.x >> 3) * blockDim.x + threadIdx.x;
  int c, chunk, res, ys, b;
  bool live = true;
  if (part.mode == 0) {
    const int c_lo = (xcd * part.nu / 8) * part.unit;
    int c_hi = ((xcd + 1) * part.nu / 8) * part.unit;
    c_hi = c_hi < c4n ? c_hi : c4n;
    const int wc = c_hi - c_lo;
    live = wc > 0 && local < static_cast<unsigned>(part.runs) * wc;
    const FastDiv& dw = wc == static_cast<int>(part.dwc[0].d) ? part.dwc[0]
                      : wc == static_cast<int>(part.dwc[1].d) ? part.dwc[1] : part.dwc[2];
    unsigned rest = fdiv(local, dw);
    c = (c_lo + static_cast<int>(local - rest * wc)) * 4;
    unsigned q = fdiv(rest, part.dchunk);
    chunk = static_cast<int>(rest - q * nchunk); rest = q;
    q = fdiv(rest, part.dres);
    res = static_cast<int>(rest - q * nres); rest = q;
    q = fdiv(rest, part.dho);
    ys = static_cast<int>(rest - q * nrows);
    b = static_cast<int>(q);
  } else {
    const int r_lo = static_cast<int>(static_cast<int64_t>(xcd) * part.rows / 8);
    const int nrow = static_cast<int>(static_cast<int64_t>(xcd + 1) * part.rows / 8) - r_lo;
    live = local < static_cast<unsigned>(nrow) * nres * nchunk * c4n;
    unsigned rest = fdiv(local, part.dc4n);
    c = static_cast<int>(local - rest * c4n) * 4;
    unsigned q = fdiv(rest, part.dchunk);
    chunk = static_cast<int>(rest - q * nchunk); rest = q;
    q = fdiv(rest, part.dres);
    res = static_cast<int>(rest - q * nres); rest = q;
    const unsigned row = r_lo + rest;
    q = fdiv(row, part.dho);
    ys = static_cast<int>(row - q * nrows);
    b = static_cast<int>(q);
  }
  const int r = p.rate;
  int y = ys;
  if (ROWS == 2) {                              // slot -> first row of the pair
    const unsigned g = fdiv(static_cast<unsigned>(ys), part.drate);
    y = static_cast<int>(2 * r * g + (ys - g * r));
  }
  live = live && res + chunk * (EPOS_DW_REP * L) * r < p.Wo && y < p.Ho;
  // A wave without any live lane leaves. In a wave that keeps some, the dead lanes stay (the
  // scale reduction below is wave-wide): they walk the run at the image's origin -- valid
  // addresses, never "interior" -- and store nothing.
  if (__builtin_amdgcn_ballot_w64(live) == 0) return;
  if (!live) {
    c = part.mode == 0 ? (xcd * part.nu / 8) * part.unit * 4 : 0;
    chunk = 0; res = 0; y = 0; b = 0;
  }
  const bool row1 = ROWS == 2 && y + r < p.Ho;   // the pair's second row exists
  float4 w[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) w[i] = ld4(p.w9c + i * p.C + c);
  const float4 bias = ld4(p.bias + c);
  // 32-bit element offsets from one per-thread base (the host checks the range)
  const float* xb = p.X + static_cast<int64_t>(b) * p.Hi * p.Wi * p.ldx + c;
  float* yb = p.Y + ((static_cast<int64_t>(b) * p.Ho + y) * p.Wo) * p.ldy + c;
  const int ldx = static_cast<int>(p.ldx), ldy = static_cast<int>(p.ldy);
  const int rowpitch = p.Wi * ldx;
  const unsigned yrow1 = static_cast<unsigned>(r * p.Wo * ldy);   // output row y + rate
  // EPOS_DW_REP consecutive runs per thread share the weight vectors
#pragma unroll 1
  for (int rep = 0; rep < EPOS_DW_REP; ++rep) {
  const int x0 = res + (chunk * EPOS_DW_REP + rep) * L * r;
  if (x0 >= p.Wo) break;
  // every tap of every output of this run inside the image?
  const bool interior = y - r >= 0 && y + (ROWS == 2 ? 2 : 1) * r < p.Hi && x0 - r >= 0 &&
                        x0 + L * r < p.Wi && (ROWS == 1 || row1);
  float4 col[L + 2][NR];
  if (__builtin_amdgcn_ballot_w64(!interior) == 0) {
    // ---- interior wave: no clamp, no select -----------------------------------
    const unsigned o00 = (y - r) * rowpitch + (x0 - r) * ldx;
    const unsigned rstep = r * rowpitch, cstep = r * ldx;
#pragma unroll
    for (int i = 0; i < L + 2; ++i)
#pragma unroll
      for (int ky = 0; ky < NR; ++ky) col[i][ky] = ld4(xb + (o00 + ky * rstep + i * cstep));
    __builtin_amdgcn_sched_barrier(0);
    dw_scale_finish();             // behind the input loads: waits for the slot word only
    __builtin_amdgcn_sched_barrier(0);
    if (RELU_IN) {
#pragma unroll
      for (int i = 0; i < L + 2; ++i)
#pragma unroll
        for (int ky = 0; ky < NR; ++ky) col[i][ky] = relu4_1op(col[i][ky]);
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
#pragma unroll
      for (int j = 0; j < L; ++j) {
        float4 acc = bias;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          acc = fma4(col[j][ky + rr], w[ky * 3 + 0], acc);
          acc = fma4(col[j + 1][ky + rr], w[ky * 3 + 1], acc);
          acc = fma4(col[j + 2][ky + rr], w[ky * 3 + 2], acc);
        }
        if (RELU_OUT) acc = relu4_1op(acc);
        st4_any(yb + (static_cast<unsigned>((x0 + j * r) * ldy) + (rr ? yrow1 : 0u)), acc, h2, hs);
      }
    }
    continue;
  }
  // ---- border wave: clamped addresses, zero padding by select ------------------
  unsigned rowoff[NR];
  bool rowok[NR];
#pragma unroll
  for (int ky = 0; ky < NR; ++ky) {
    const int yi = y + (ky - 1) * r;
    rowok[ky] = yi >= 0 && yi < p.Hi;
    rowoff[ky] = (rowok[ky] ? yi : 0) * rowpitch;
  }
#pragma unroll
  for (int i = 0; i < L + 2; ++i) {
    const int xi = x0 + (i - 1) * r;
    const bool xok = xi >= 0 && xi < p.Wi;
    const unsigned off = (xok ? xi : 0) * ldx;
#pragma unroll
    for (int ky = 0; ky < NR; ++ky) col[i][ky] = ld4(xb + (rowoff[ky] + off));
  }
  __builtin_amdgcn_sched_barrier(0);
  dw_scale_finish();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < L + 2; ++i) {
    const int xi = x0 + (i - 1) * r;
    const bool xok = xi >= 0 && xi < p.Wi;
#pragma unroll
    for (int ky = 0; ky < NR; ++ky) {
      float4 v = col[i][ky];
      if (!(xok && rowok[ky])) v = make_float4(0.f, 0.f, 0.f, 0.f);
      col[i][ky] = RELU_IN ? relu4_1op(v) : v;
    }
  }
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
    if (rr == 1 && !row1) break;
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const int x = x0 + j * r;
      float4 acc = bias;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        acc = fma4(col[j][ky + rr], w[ky * 3 + 0], acc);
        acc = fma4(col[j + 1][ky + rr], w[ky * 3 + 1], acc);
        acc = fma4(col[j + 2][ky + rr], w[ky * 3 + 2], acc);
      }
      if (RELU_OUT) acc = relu4_1op(acc);
      if (x < p.Wo && live) st4_any(yb + (static_cast<unsigned>(x * ldy) + (rr ? yrow1 : 0u)), acc, h2, hs);
    }
  }
  }   // rep
}

// --------------------------------------------------------------------------
// im2col for dense 3x3 convs (scalar: C may be 3).
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col3x3_kernel(EposIm2colArgs p,
                                                        int64_t total) {
  const int64_t id = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p.amax_clear && id < p.amax_words) p.amax_clear[id] = 0u;    // the plan's slot table
  if (id >= total) return;
  const int col = static_cast<int>(id % p.ldcol);
  int64_t pix = id / p.ldcol;
  float v = 0.f;
  if (col < 9 * p.C) {
    const int tap = col / p.C, c = col - tap * p.C;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int xo = static_cast<int>(pix % p.Wo);
    const int64_t t = pix / p.Wo;
    const int yo = static_cast<int>(t % p.Ho);
    const int b = static_cast<int>(t / p.Ho);
    const int yi = yo * p.stride - p.pad + ky * p.rate;
    const int xi = xo * p.stride - p.pad + kx * p.rate;
    if (yi >= 0 && yi < p.Hi && xi >= 0 && xi < p.Wi) {
      v = p.X[((static_cast<int64_t>(b) * p.Hi + yi) * p.Wi + xi) * p.ldx + c];
      if (p.preprocess) v = (2.0f / 255.0f) * v - 1.0f;   // feature.py:171-174
    }
  }
  p.col[pix * p.ldcol + col] = v;
}

// --------------------------------------------------------------------------
// Global average pool: block = (image, 64-channel group); 16 float4 lanes x 64
// row phases; fixed-order LDS reduction (deterministic).
// --------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void global_avg_pool_kernel(
    const float* X, int64_t ldx, float* Y, int HW, int C) {
  __shared__ float4 part[64][16];
  const int b = blockIdx.y;
  const int c = blockIdx.x * 64 + (threadIdx.x & 15) * 4;
  const int phase = threadIdx.x >> 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    const float* xb = X + static_cast<int64_t>(b) * HW * ldx + c;
    int r = phase;
    for (; r + 7 * 64 < HW; r += 8 * 64) {        // eight rows in flight per thread
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld4(xb + static_cast<int64_t>(r + u * 64) * ldx);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; r < HW; r += 64) {
      const float4 v = ld4(xb + static_cast<int64_t>(r) * ldx);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[phase][threadIdx.x & 15] = s;
  __syncthreads();
  if (phase == 0 && c < C) {
    float4 t = part[0][threadIdx.x];
    for (int i = 1; i < 64; ++i) {
      const float4 v = part[i][threadIdx.x];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    const float n = static_cast<float>(HW);
    st4(Y + static_cast<int64_t>(b) * C + c,
        make_float4(t.x / n, t.y / n, t.z / n, t.w / n));
  }
}

// --------------------------------------------------------------------------
// Bilinear resize, align_corners=True (TF resize_bilinear arithmetic order).
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_bilinear_kernel(
    const float* X, int64_t ldx, float* Y, int64_t ldy, int Hi, int Wi, int Ho,
    int Wo, int c4n, float sy, float sx, int64_t total) {
  const int64_t id = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int c = static_cast<int>(id % c4n) * 4;
  int64_t pix = id / c4n;
  const int xo = static_cast<int>(pix % Wo);
  pix /= Wo;
  const int yo = static_cast<int>(pix % Ho);
  const int b = static_cast<int>(pix / Ho);
  const float fy = yo * sy, fx = xo * sx;
  const int y0 = static_cast<int>(floorf(fy)), x0 = static_cast<int>(floorf(fx));
  const int y1 = min(static_cast<int>(ceilf(fy)), Hi - 1);
  const int x1 = min(static_cast<int>(ceilf(fx)), Wi - 1);
  const float ly = fy - y0, lx = fx - x0;
  const float* xb = X + static_cast<int64_t>(b) * Hi * Wi * ldx + c;
  const float4 tl = ld4(xb + (static_cast<int64_t>(y0) * Wi + x0) * ldx);
  const float4 tr = ld4(xb + (static_cast<int64_t>(y0) * Wi + x1) * ldx);
  const float4 bl = ld4(xb + (static_cast<int64_t>(y1) * Wi + x0) * ldx);
  const float4 br = ld4(xb + (static_cast<int64_t>(y1) * Wi + x1) * ldx);
  auto lerp = [](float a, float bb, float w) { return a + (bb - a) * w; };
  float4 o;
  o.x = lerp(lerp(tl.x, tr.x, lx), lerp(bl.x, br.x, lx), ly);
  o.y = lerp(lerp(tl.y, tr.y, lx), lerp(bl.y, br.y, lx), ly);
  o.z = lerp(lerp(tl.z, tr.z, lx), lerp(bl.z, br.z, lx), ly);
  o.w = lerp(lerp(tl.w, tr.w, lx), lerp(bl.w, br.w, lx), ly);
  st4(Y + ((static_cast<int64_t>(b) * Ho + yo) * Wo + xo) * ldy + c, o);
}

// --------------------------------------------------------------------------
// Softmax over groups of G <= 64 consecutive floats; one wave per group.
// --------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void softmax_groups_kernel(float* X,
                                                             int64_t n_groups,
                                                             int G) {
  const int lane = threadIdx.x & 63;
  const int64_t g = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (g >= n_groups) return;
  float* x = X + g * G;
  const bool on = lane < G;
  const float v = on ? x[lane] : -INFINITY;
  const float m = wave_max(v);
  const float e = on ? expf(v - m) : 0.f;
  const float s = wave_sum(e);
  if (on) x[lane] = e / s;
}

// G == 64 (the fragment axis): a wave handles FOUR groups, 16 lanes x float4 each --
// a quarter of the waves and fewer shuffle levels than the one-float-per-lane form. The same arithmetic (max / sum: in-lane pairs first, then the 8-4-2-1 xor
// butterfly over the 16 lanes) is used by softmax_slots64_kernel, so dense and
// sparse-head runs stay bit-identical.
// (softmax64_lane16: h2_scale.h)
// The dense fragment-confidence head is 103 MB at C2: read once, written once, re-read only
// sparsely by the correspondence stage -- streaming (non-temporal) accesses keep it from
// sweeping the Infinity Cache (EPOS_SOFTMAX_NT=0 at build time restores plain accesses).
#ifndef EPOS_SOFTMAX_NT
#define EPOS_SOFTMAX_NT 1
#endif
typedef float sm_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void softmax_groups64_kernel(float* X, int64_t n_groups) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const bool on = g < n_groups;
  float* x = X + (on ? g : 0) * 64 + (lane & 15) * 4;
#if EPOS_SOFTMAX_NT
  const sm_f32x4 in = __builtin_nontemporal_load(reinterpret_cast<const sm_f32x4*>(x));
  const float4 r = softmax64_lane16(make_float4(in[0], in[1], in[2], in[3]));
  if (on) {
    const sm_f32x4 o = {r.x, r.y, r.z, r.w};
    __builtin_nontemporal_store(o, reinterpret_cast<sm_f32x4*>(x));
  }
#else
  const float4 r = softmax64_lane16(ld4(x));             // shuffles: all lanes take part
  if (on) st4(x, r);
#endif
}

__global__ __launch_bounds__(256) void softmax_slots64_kernel(
    float* X, const EposCorrSlot* __restrict__ slots, int P, int O) {
  const int lane = threadIdx.x & 63;
  const int p = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const int s = blockIdx.y;
  const int img = slots[s].image, obj = slots[s].obj_id;
  const bool on = p < P;
  float* x = X + ((static_cast<int64_t>(img) * P + (on ? p : 0)) * O + (obj - 1)) * 64 +
             (lane & 15) * 4;
  const float4 r = softmax64_lane16(ld4(x));
  if (on) st4(x, r);
}

// Softmax over the F fragment confidences of the given (image, object) slots only
// (sparse-head mode): group of slot s, pixel p at X + ((img*P + p)*O + obj-1)*F.
__global__ __launch_bounds__(256) void softmax_slots_kernel(
    float* X, const EposCorrSlot* __restrict__ slots, int P, int O, int F) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  const int s = blockIdx.y;
  const int img = slots[s].image, obj = slots[s].obj_id;
  float* x = X + ((static_cast<int64_t>(img) * P + p) * O + (obj - 1)) * F;
  const bool on = lane < F;
  const float v = on ? x[lane] : -INFINITY;
  const float m = wave_max(v);
  const float e = on ? expf(v - m) : 0.f;
  const float sum = wave_sum(e);
  if (on) x[lane] = e / sum;
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* X, int64_t ldx,
                                                     int64_t* labels, int64_t P,
                                                     int C) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float* x = X + i * ldx;
  float best = x[0];
  int arg = 0;
  for (int c = 1; c < C; ++c) {
    const float v = x[c];
    if (v > best) { best = v; arg = c; }
  }
  labels[i] = arg;
}

// Sparse update of a device tensor from host-prepared data: block b of `width` floats goes to
// dst + offsets[b]. (bench.py --planted-poses writes rendered head values for the target
// objects over the network's outputs with it -- three launches per image.)
__global__ __launch_bounds__(256) void scatter_blocks_kernel(float* __restrict__ dst,
                                                             const int64_t* __restrict__ offsets,
                                                             const float* __restrict__ src,
                                                             int64_t total, int width) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t b = i / width;
  dst[offsets[b] + (i - b * width)] = src[i];
}

// Decoded frames -> the network's input tensor: tf.cast(decode_image(...), tf.float32)
// (datagen.py:435-436) on the device, so that a frame crosses PCIe as 1 byte per value
// (0.92 MB instead of 3.7 MB at 640 x 480). 16 values per thread: one dwordx4 load, four
// float4 stores; exact (every uint8 is an fp32 value).
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ X,
                                                        float* __restrict__ Y, int64_t n) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 16;
  if (i >= n) return;
  if (i + 16 <= n) {
    const uint4 v = *reinterpret_cast<const uint4*>(X + i);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      reinterpret_cast<float4*>(Y + i)[q] = make_float4(
          static_cast<float>(w[q] & 0xffu), static_cast<float>((w[q] >> 8) & 0xffu),
          static_cast<float>((w[q] >> 16) & 0xffu), static_cast<float>(w[q] >> 24));
  } else {
    for (int64_t j = i; j < n; ++j) Y[j] = static_cast<float>(X[j]);
  }
}

// --------------------------------------------------------------------------
// ResNet-v1-beta helpers (BASELINE config C5): 3x3 stride-2 'SAME' max pool
// (net_resnet_v1_beta.py:190), spatial subsampling (slim resnet_utils.subsample,
// the identity shortcut of a strided unit) and the unit's final relu(a + b)
// where the pre-activation sum is also an end point.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool3x3_s2_kernel(
    const float* X, int64_t ldx, float* Y, int64_t ldy, int Hi, int Wi, int Ho,
    int Wo, int c4n, int pad_y, int pad_x, int64_t total) {
  const int64_t id = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int c = static_cast<int>(id % c4n) * 4;
  int64_t pix = id / c4n;
  const int xo = static_cast<int>(pix % Wo);
  pix /= Wo;
  const int yo = static_cast<int>(pix % Ho);
  const int b = static_cast<int>(pix / Ho);
  const float* xb = X + static_cast<int64_t>(b) * Hi * Wi * ldx + c;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yi = yo * 2 - pad_y + ky;
    if (yi < 0 || yi >= Hi) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xi = xo * 2 - pad_x + kx;
      if (xi < 0 || xi >= Wi) continue;
      const float4 v = ld4(xb + (static_cast<int64_t>(yi) * Wi + xi) * ldx);
      m = make_float4(fmaxf(m.x, v.x), fmaxf(m.y, v.y), fmaxf(m.z, v.z),
                      fmaxf(m.w, v.w));
    }
  }
  st4(Y + ((static_cast<int64_t>(b) * Ho + yo) * Wo + xo) * ldy + c, m);
}

__global__ __launch_bounds__(256) void subsample_kernel(
    const float* X, int64_t ldx, float* Y, int64_t ldy, int Hi, int Wi, int Ho,
    int Wo, int c4n, int factor, int64_t total) {
  const int64_t id = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int c = static_cast<int>(id % c4n) * 4;
  int64_t pix = id / c4n;
  const int xo = static_cast<int>(pix % Wo);
  pix /= Wo;
  const int yo = static_cast<int>(pix % Ho);
  const int b = static_cast<int>(pix / Ho);
  st4(Y + ((static_cast<int64_t>(b) * Ho + yo) * Wo + xo) * ldy + c,
      ld4(X + ((static_cast<int64_t>(b) * Hi + yo * factor) * Wi + xo * factor) * ldx + c));
}

__global__ __launch_bounds__(256) void add_relu_kernel(const float* A, const float* B,
                                                       float* Y, int64_t n4) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 a = ld4(A + 4 * i), b = ld4(B + 4 * i);
  st4(Y + 4 * i, relu4(make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w)));
}

inline unsigned blocks_for(int64_t total, int threads) {
  return static_cast<unsigned>(ceil_div(total, threads));
}

}  // namespace
}  // namespace epos

using namespace epos;


extern "C" int epos_depthwise3x3_f32(const EposDepthwiseArgs* a, void* stream) {
  EPOS_REQUIRE(a && a->X && a->w9c && a->bias && a->Y, "null pointer");
  EPOS_REQUIRE(a->C % 4 == 0 && a->ldx % 4 == 0 && a->ldy % 4 == 0,
               "C, ldx, ldy must be multiples of 4");
  EPOS_REQUIRE(a->stride == 1 || (a->stride == 2 && a->rate == 1),
               "stride 2 requires rate 1");
  EPOS_REQUIRE(!a->y_h2 || a->x_amax, "y_h2 needs the absmax slot of X (x_amax)");
  const int c4n = a->C / 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a->stride == 1 && a->Hi == a->Ho && a->Wi == a->Wo) {
#ifndef EPOS_DW_L
#define EPOS_DW_L 4
#endif
    constexpr int L = EPOS_DW_L;
    static const int threads = [] {        // EPOS_DW_THREADS=64|128|256 (tuning)
      const char* e = getenv("EPOS_DW_THREADS");
      return e ? atoi(e) : 256;
    }();
    const int nres = a->rate < a->Wo ? a->rate : a->Wo;
    const int per_res = static_cast<int>(ceil_div(a->Wo, a->rate));
    const int nchunk = static_cast<int>(ceil_div(per_res, L * EPOS_DW_REP));
    // two output rows per thread (EPOS_DW_ROWS=1 keeps one: the A/B switch)
    static const int rows_env = [] {
      const char* e = getenv("EPOS_DW_ROWS");
      return e ? atoi(e) : 2;
    }();
    // ... unless the launch would then not even give every CU one workgroup (the 60 x 80 x
    // 256 tensor: 4.8 us with one row per thread, 6.7 with two)
    const int64_t threads2 = static_cast<int64_t>(a->B) * ceil_div(a->Ho, 2) *
                             ceil_div(a->Wo, L) * c4n;
    const int ROWS = (rows_env == 1 || (rows_env != 22 && threads2 < 256 * 256)) ? 1 : 2;
    const int nrows = ROWS == 2 ? static_cast<int>(ceil_div(a->Ho, 2 * a->rate)) * a->rate : a->Ho;
    const int64_t runs = static_cast<int64_t>(a->B) * nrows * nres * nchunk;
    if (runs == 0) return EPOS_OK;
    EPOS_REQUIRE(runs * c4n < (1LL << 31) &&
                 static_cast<int64_t>(a->Hi) * a->Wi * a->ldx < (1LL << 29) &&
                 static_cast<int64_t>(a->Ho) * a->Wo * a->ldy < (1LL << 29),
                 "tensor too large for 32-bit offsets within one image");
    static const int force_mode = [] {     // EPOS_DW_MODE=0|1 (tuning)
      const char* e = getenv("EPOS_DW_MODE");
      return e ? atoi(e) : -1;
    }();
    auto fast_div = [](unsigned d) {
      FastDiv f; f.d = d;
      unsigned l = 0;
      while ((1ull << l) < d) ++l;
      f.mul = static_cast<unsigned>(((1ull << 32) * ((1ull << l) - d)) / d + 1);
      f.sh1 = l > 0 ? 1 : 0;
      f.sh2 = l > 0 ? l - 1 : 0;
      return f;
    };
    DwPartition part;
    // channel slices when a slice is at least one 128-byte line wide and aligned
    // (or wide enough that the two ragged lines per pixel do not matter)
    part.mode = (c4n >= 128 || (c4n >= 64 && c4n % 8 == 0)) ? 0 : 1;
    if (force_mode >= 0) part.mode = force_mode;
    if (c4n < 8) part.mode = 1;
    part.runs = static_cast<int>(runs);
    part.rows = a->B * nrows;
    // Slices that start on 128-byte lines when every pixel's channel vector does (rows of a
    // multiple of 32 floats from line-aligned bases; the plan pads 728 -> 736): no line is then
    // fetched by two XCDs. Otherwise slices of float4 granularity, as balanced as they get (a
    // 728-float row has 7 slice boundaries inside lines: ~30 % of the input read twice).
    const bool lines = a->ldx % 32 == 0 && a->ldy % 32 == 0 &&
                       (reinterpret_cast<uintptr_t>(a->X) & 127) == 0 &&
                       (reinterpret_cast<uintptr_t>(a->Y) & 127) == 0 && c4n >= 64;
    part.unit = lines ? 8 : 1;
    part.nu = static_cast<int>(ceil_div(c4n, part.unit));
    int widths[3] = {0, 0, 0}, nwid = 0, wmax = 0;
    for (int x = 0; x < 8; ++x) {
      const int lo = (x * part.nu / 8) * part.unit;
      int hi = ((x + 1) * part.nu / 8) * part.unit;
      hi = hi < c4n ? hi : c4n;
      const int wdt = hi - lo;
      if (wdt <= 0) continue;
      wmax = wdt > wmax ? wdt : wmax;
      bool seen = false;
      for (int i = 0; i < nwid; ++i) seen = seen || widths[i] == wdt;
      if (!seen && nwid < 3) widths[nwid++] = wdt;
    }
    for (int i = 0; i < 3; ++i) part.dwc[i] = fast_div(widths[i] > 0 ? widths[i] : 1);
    part.dc4n = fast_div(c4n);
    part.dchunk = fast_div(nchunk);
    part.dres = fast_div(nres);
    part.dho = fast_div(nrows);
    part.drate = fast_div(a->rate);
    int64_t per_xcd;                        // items of the busiest XCD
    if (part.mode == 0) per_xcd = runs * (wmax > 0 ? wmax : 1);
    else per_xcd = ceil_div(part.rows, 8) * nres * nchunk * c4n;
    const unsigned grid = 8 * blocks_for(per_xcd, threads);
    const int v = (a->relu_in ? 2 : 0) | (a->relu_out ? 1 : 0);
    // EPOS_DW_LDS_KB=n: reserve n KB of (unused) LDS per workgroup -- an occupancy throttle
    // for co-residency experiments with the GEMM's 80 KB workgroups
    static const unsigned lds_pad = [] {
      const char* e = getenv("EPOS_DW_LDS_KB");
      return e ? static_cast<unsigned>(atoi(e)) * 1024u : 0u;
    }();
#define EPOS_DW_LAUNCH(RI, RO, RW)                                                          \
    hipLaunchKernelGGL((depthwise3x3_s1_kernel<L, RI, RO, RW>), dim3(grid), dim3(threads),   \
                       lds_pad, st, *a, c4n, nres, nchunk, nrows, part)
    if (ROWS == 2) {
      if (v == 0) EPOS_DW_LAUNCH(false, false, 2);
      else if (v == 1) EPOS_DW_LAUNCH(false, true, 2);
      else if (v == 2) EPOS_DW_LAUNCH(true, false, 2);
      else EPOS_DW_LAUNCH(true, true, 2);
    } else {
      if (v == 0) EPOS_DW_LAUNCH(false, false, 1);
      else if (v == 1) EPOS_DW_LAUNCH(false, true, 1);
      else if (v == 2) EPOS_DW_LAUNCH(true, false, 1);
      else EPOS_DW_LAUNCH(true, true, 1);
    }
#undef EPOS_DW_LAUNCH
    return launch_status("depthwise3x3_s1_kernel");
  }
  const int64_t total = static_cast<int64_t>(a->B) * a->Ho * a->Wo * c4n;
  if (total == 0) return EPOS_OK;
  hipLaunchKernelGGL(depthwise3x3_kernel, dim3(blocks_for(total, 256)), dim3(256),
                     0, st, *a, c4n, total);
  return launch_status("depthwise3x3_kernel");
}

extern "C" int epos_im2col3x3_f32(const EposIm2colArgs* a, void* stream) {
  EPOS_REQUIRE(a && a->X && a->col, "null pointer");
  EPOS_REQUIRE(a->ldcol >= 9 * a->C, "ldcol too small");
  const int64_t total = static_cast<int64_t>(a->B) * a->Ho * a->Wo * a->ldcol;
  EPOS_REQUIRE(!a->amax_clear || (a->amax_words >= 0 && a->amax_words <= total),
               "amax_words exceeds the launch's thread count");
  if (total == 0) return EPOS_OK;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(blocks_for(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), *a, total);
  return launch_status("im2col3x3_kernel");
}

extern "C" int epos_global_avg_pool_f32(const float* X, int64_t ldx, float* Y,
                                        int B, int HW, int C, void* stream) {
  EPOS_REQUIRE(X && Y, "null pointer");
  EPOS_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && HW > 0, "C, ldx multiples of 4");
  hipLaunchKernelGGL(global_avg_pool_kernel,
                     dim3(static_cast<unsigned>(ceil_div(C, 64)), B), dim3(1024),
                     0, static_cast<hipStream_t>(stream), X, ldx, Y, HW, C);
  return launch_status("global_avg_pool_kernel");
}

namespace epos { namespace {
// Y[b, c..c+3] = (sum over the blocks of image b of P[b * blocks + i, c..c+3]) / hw, blocks
// in order: the second half of the image-pooling mean whose first half an fp16-pair GEMM's
// epilogue wrote as 32-row block sums (EposPointwiseArgs.col_sums).
__global__ __launch_bounds__(256) void pool_partial_kernel(const float* P, int64_t ldp,
                                                           float* Y, int blocks, int C, int hw) {
  // 16 row groups x 16 channel quads per workgroup: a thread adds every 16th block row (the
  // loads of a thread are independent, so the ~10 of them for 150 rows overlap instead of
  // queueing behind each other as in a one-thread-per-column loop: 39 -> ~6 us), then the 16
  // partial sums meet in LDS in a fixed order.
  __shared__ float4 red[16][16];
  const int b = blockIdx.y;
  const int q = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int c = (blockIdx.x * 16 + q) * 4;
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    const float* pb = P + static_cast<int64_t>(b) * blocks * ldp + c;
#pragma unroll 4
    for (int i = r; i < blocks; i += 16) {
      const float4 v = ld4(pb + static_cast<int64_t>(i) * ldp);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
  }
  red[r][q] = t;
  __syncthreads();
  if (r != 0 || c >= C) return;
  for (int i = 1; i < 16; ++i) {
    const float4 v = red[i][q];
    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
  }
  const float n = static_cast<float>(hw);
  st4(Y + static_cast<int64_t>(b) * C + c, make_float4(t.x / n, t.y / n, t.z / n, t.w / n));
}
} }

extern "C" int epos_global_avg_pool_partial_f32(const float* P, int64_t ldp, float* Y,
                                                int32_t B, int32_t blocks, int32_t C,
                                                int32_t hw, void* stream) {
  EPOS_REQUIRE(P && Y, "null pointer");
  EPOS_REQUIRE(C % 4 == 0 && ldp % 4 == 0 && ldp >= C && blocks > 0 && hw > 0 && B > 0,
               "C, ldp multiples of 4; blocks, hw, B > 0");
  hipLaunchKernelGGL(epos::pool_partial_kernel,
                     dim3(static_cast<unsigned>(epos::ceil_div(C, 64)), B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), P, ldp, Y, blocks, C, hw);
  return epos::launch_status("pool_partial_kernel");
}

extern "C" int epos_resize_bilinear_f32(const float* X, int64_t ldx, float* Y,
                                        int64_t ldy, int B, int Hi, int Wi,
                                        int Ho, int Wo, int C, void* stream) {
  EPOS_REQUIRE(X && Y, "null pointer");
  EPOS_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "multiples of 4");
  const float sy = Ho > 1 ? static_cast<float>(Hi - 1) / (Ho - 1) : 0.f;
  const float sx = Wo > 1 ? static_cast<float>(Wi - 1) / (Wo - 1) : 0.f;
  const int c4n = C / 4;
  const int64_t total = static_cast<int64_t>(B) * Ho * Wo * c4n;
  if (total == 0) return EPOS_OK;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(blocks_for(total, 256)),
                     dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx, Y,
                     ldy, Hi, Wi, Ho, Wo, c4n, sy, sx, total);
  return launch_status("resize_bilinear_kernel");
}

extern "C" int epos_softmax_groups_f32(float* X, int64_t n_groups, int G,
                                       void* stream) {
  EPOS_REQUIRE(X, "null pointer");
  EPOS_REQUIRE(G >= 1 && G <= 64, "G must be in [1, 64]");
  if (n_groups == 0) return EPOS_OK;
  if (G == 64 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
    hipLaunchKernelGGL(softmax_groups64_kernel, dim3(blocks_for(n_groups, 16)), dim3(256),
                       0, static_cast<hipStream_t>(stream), X, n_groups);
    return launch_status("softmax_groups64_kernel");
  }
  hipLaunchKernelGGL(softmax_groups_kernel, dim3(blocks_for(n_groups, 4)),
                     dim3(256), 0, static_cast<hipStream_t>(stream), X, n_groups,
                     G);
  return launch_status("softmax_groups_kernel");
}

extern "C" int epos_argmax_i64(const float* X, int64_t ldx, int64_t* labels,
                               int64_t P, int C, void* stream) {
  EPOS_REQUIRE(X && labels, "null pointer");
  if (P == 0) return EPOS_OK;
  hipLaunchKernelGGL(argmax_kernel, dim3(blocks_for(P, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), X, ldx, labels, P, C);
  return launch_status("argmax_kernel");
}

extern "C" int epos_scatter_blocks_f32(float* dst, const int64_t* offsets, const float* src,
                                       int64_t n_blocks, int width, void* stream) {
  EPOS_REQUIRE(n_blocks >= 0 && width > 0, "bad sizes");
  if (n_blocks == 0) return EPOS_OK;
  EPOS_REQUIRE(dst && offsets && src, "null pointer");
  const int64_t total = n_blocks * width;
  hipLaunchKernelGGL(scatter_blocks_kernel, dim3(blocks_for(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), dst, offsets, src, total, width);
  return launch_status("scatter_blocks_kernel");
}

extern "C" int epos_u8_to_f32(const uint8_t* X, float* Y, int64_t n, void* stream) {
  EPOS_REQUIRE(X && Y, "null pointer");
  EPOS_REQUIRE(reinterpret_cast<uintptr_t>(X) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(Y) % 16 == 0, "16-byte aligned buffers");
  if (n == 0) return EPOS_OK;
  hipLaunchKernelGGL(u8_to_f32_kernel, dim3(blocks_for((n + 15) / 16, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), X, Y, n);
  return launch_status("u8_to_f32_kernel");
}

extern "C" int epos_maxpool3x3_s2_f32(const float* X, int64_t ldx, float* Y,
                                      int64_t ldy, int B, int Hi, int Wi, int C,
                                      void* stream) {
  EPOS_REQUIRE(X && Y, "null pointer");
  EPOS_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "multiples of 4");
  const int Ho = (Hi + 1) / 2, Wo = (Wi + 1) / 2;          // TF 'SAME'
  const int ty = (Ho - 1) * 2 + 3 - Hi, tx = (Wo - 1) * 2 + 3 - Wi;
  const int pad_y = ty > 0 ? ty / 2 : 0, pad_x = tx > 0 ? tx / 2 : 0;
  const int c4n = C / 4;
  const int64_t total = static_cast<int64_t>(B) * Ho * Wo * c4n;
  if (total == 0) return EPOS_OK;
  hipLaunchKernelGGL(maxpool3x3_s2_kernel, dim3(blocks_for(total, 256)), dim3(256),
                     0, static_cast<hipStream_t>(stream), X, ldx, Y, ldy, Hi, Wi,
                     Ho, Wo, c4n, pad_y, pad_x, total);
  return launch_status("maxpool3x3_s2_kernel");
}

extern "C" int epos_subsample_f32(const float* X, int64_t ldx, float* Y,
                                  int64_t ldy, int B, int Hi, int Wi, int C,
                                  int factor, void* stream) {
  EPOS_REQUIRE(X && Y, "null pointer");
  EPOS_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && factor >= 1,
               "multiples of 4");
  const int Ho = (Hi - 1) / factor + 1, Wo = (Wi - 1) / factor + 1;
  const int c4n = C / 4;
  const int64_t total = static_cast<int64_t>(B) * Ho * Wo * c4n;
  if (total == 0) return EPOS_OK;
  hipLaunchKernelGGL(subsample_kernel, dim3(blocks_for(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), X, ldx, Y, ldy, Hi, Wi, Ho,
                     Wo, c4n, factor, total);
  return launch_status("subsample_kernel");
}

extern "C" int epos_add_relu_f32(const float* A, const float* B, float* Y,
                                 int64_t n, void* stream) {
  EPOS_REQUIRE(A && B && Y, "null pointer");
  EPOS_REQUIRE(n % 4 == 0, "n must be a multiple of 4");
  if (n == 0) return EPOS_OK;
  hipLaunchKernelGGL(add_relu_kernel, dim3(blocks_for(n / 4, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), A, B, Y, n / 4);
  return launch_status("add_relu_kernel");
}

extern "C" int epos_softmax_slots_f32(float* X, const EposCorrSlot* slots, int S,
                                      int P, int O, int F, void* stream) {
  EPOS_REQUIRE(X && slots, "null pointer");
  EPOS_REQUIRE(F >= 1 && F <= 64, "F must be in [1, 64]");
  if (S == 0 || P == 0) return EPOS_OK;
  if (F == 64 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
    hipLaunchKernelGGL(softmax_slots64_kernel, dim3(blocks_for(P, 16), S), dim3(256), 0,
                       static_cast<hipStream_t>(stream), X, slots, P, O);
    return launch_status("softmax_slots64_kernel");
  }
  hipLaunchKernelGGL(softmax_slots_kernel, dim3(blocks_for(P, 4), S), dim3(256), 0,
                     static_cast<hipStream_t>(stream), X, slots, P, O, F);
  return launch_status("softmax_slots_kernel");
}
