// fp16-pair representation shared by the depthwise kernels (which can write their output
// already split) and the fp16-pair GEMM (pointwise_gemm_h2.hip): the power-of-two scale of
// a tensor from the absmax slot(s) that bound it, and the split of scaled values.
#pragma once
#include "common.h"

namespace epos {
namespace {

typedef _Float16 h2_f16x2 __attribute__((ext_vector_type(2)));
typedef float h2_f32x2 __attribute__((ext_vector_type(2)));

// Scale of a tensor from the bound in its absmax slot(s) (include/epos_hip.h): a power of
// two s with s * bound in [2^14, 2^15) (fp16 overflows at 65520), and its inverse. bound =
// gain * max(slot, slot2) + bias (gain == 0 reads as 1, 0). A non-finite bound (an Inf / NaN
// upstream) gives s = 1: such rows come out non-finite, the others right. Wave-wide; every
// lane returns the same values.
// The two halves of h2_scale: the lane's word(s) of the slot(s) -- a global load whose
// latency a caller may want to overlap with other work -- and the wave-wide rest.
// (the two raw words are only combined in h2_scale_finish: nothing waits for them here)
__device__ __forceinline__ void h2_scale_load(const unsigned* amax, const unsigned* amax2,
                                              int lane, unsigned& v, unsigned& v2) {
  v = amax[lane];
  // unconditional (a load under a branch is waited for on the spot): without a second slot
  // the first one is read twice
  const unsigned* a2 = amax2 ? amax2 : amax;
  v2 = a2[lane];
}
__device__ __forceinline__ void h2_scale_finish(unsigned v, unsigned v2, float gain,
                                                float bias, float& s, float& inv);
__device__ __forceinline__ void h2_scale(const unsigned* amax, const unsigned* amax2,
                                         float gain, float bias, int lane, float& s,
                                         float& inv) {
  unsigned v, v2;
  h2_scale_load(amax, amax2, lane, v, v2);
  h2_scale_finish(v, v2, gain, bias, s, inv);
}
__device__ __forceinline__ void h2_scale_finish(unsigned v, unsigned v2, float gain,
                                                float bias, float& s, float& inv) {
  v = v2 > v ? v2 : v;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned w = __shfl_xor(v, o, 64);
    v = w > v ? w : v;
  }
  float bound = __uint_as_float(v);
  if (gain != 0.f) bound = gain * bound + bias;
  const unsigned e = __float_as_uint(bound) >> 23;            // bound >= 0: no sign bit
  int sb = 268 - static_cast<int>(e);                         // 2^(14 - (e - 127)), biased
  sb = sb > 253 ? 253 : sb;
  if (e >= 255u) sb = 127;
  s = __uint_as_float(static_cast<unsigned>(sb) << 23);
  inv = __uint_as_float(static_cast<unsigned>(254 - sb) << 23);
}

// two fp32 values -> packed (hi, hi) and (mid, mid) fp16 pairs; s = the tensor's scale:
//   t = x * s, hi = rn_fp16(t), mid = rn_fp16((t - hi) * 2^11)        (t - hi exact)
__device__ __forceinline__ void h2_split_pair(float x0, float x1, float s, unsigned& hi,
                                              unsigned& mid) {
  const h2_f32x2 t = {x0 * s, x1 * s};
  const h2_f16x2 h = __builtin_convertvector(t, h2_f16x2);        // v_cvt_pk_f16_f32 (RNE)
  const h2_f32x2 r = (t - __builtin_convertvector(h, h2_f32x2)) * 2048.f;
  const h2_f16x2 m = __builtin_convertvector(r, h2_f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  mid = __builtin_bit_cast(unsigned, m);
}

// Softmax over a group of 64 values held by 16 consecutive lanes x float4 (the fragment
// axis): max / sum by in-lane pairs first, then the 8-4-2-1 xor butterfly over the 16 lanes.
// ONE definition for the stand-alone kernels (softmax_groups64_kernel, softmax_slots64_kernel
// in layers.hip) and for the fp16-pair GEMM's epilogue (EposPointwiseArgs.softmax64), whose
// row phase holds a staged row in exactly this arrangement -- so dense, sparse-head and
// fused runs give the same bits. All 64 lanes must take part.
__device__ __forceinline__ float4 softmax64_lane16(float4 v) {
  float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
  v.x = expf(v.x - m); v.y = expf(v.y - m); v.z = expf(v.z - m); v.w = expf(v.w - m);
  float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
  return make_float4(v.x / s, v.y / s, v.z / s, v.w / s);
}

}  // namespace
}  // namespace epos
