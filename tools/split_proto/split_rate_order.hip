// Rate probe for the split-bf16 inner loop: per k-step (16 k) a wave reads its fp32 A
// fragments from LDS (RB row blocks x 2 ds_read_b128), splits them into three bf16
// pieces on the VALU, reads pre-split B fragments from LDS (CB col blocks x 3 planes)
// and issues RB*CB*6 v_mfma_f32_32x32x16_bf16. Reports MFMA-pipe utilisation
// (32 cycles per MFMA per SIMD) at 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 split_rate.hip -o split_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned pack_hi(unsigned a, unsigned b) {   // {b.hi16, a.hi16}
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}
__device__ inline void split8(const f32x4 x0, const f32x4 x1, u32x4& h, u32x4& m, u32x4& l) {
  float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  unsigned hb[8], mb[8], lb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hb[j] = __float_as_uint(x[j]);
    const float r1 = x[j] - __uint_as_float(hb[j] & 0xffff0000u);
    mb[j] = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(mb[j] & 0xffff0000u);
    lb[j] = __float_as_uint(r2);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = pack_hi(hb[2 * j], hb[2 * j + 1]);
    m[j] = pack_hi(mb[2 * j], mb[2 * j + 1]);
    l[j] = pack_hi(lb[2 * j], lb[2 * j + 1]);
  }
}

template <int RB, int CB, int SPLIT, int ORDER = 0>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters,
                                         unsigned long long* clk) {
  extern __shared__ float lds[];     // 32 KB of A-like fp32 + 24 KB of B pieces
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 14336; i += 256) lds[i] = in[(blockIdx.x * 131 + i) & 0xfffff];
  __syncthreads();
  f32x16 acc[RB * CB];
  for (int j = 0; j < RB * CB; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const f32x4* a4 = reinterpret_cast<const f32x4*>(lds);
  const u32x4* b4 = reinterpret_cast<const u32x4*>(lds + 8192);
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    const int st = (it & 3) * 512;
    u32x4 ah[RB], am[RB], al[RB], bp[CB][3];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const f32x4 x0 = a4[st + rb * 128 + lane], x1 = a4[st + rb * 128 + 64 + lane];
      if (SPLIT) split8(x0, x1, ah[rb], am[rb], al[rb]);
      else {
        for (int j = 0; j < 4; ++j) { ah[rb][j] = __float_as_uint(x0[j]); am[rb][j] = __float_as_uint(x1[j]); al[rb][j] = ah[rb][j] ^ am[rb][j]; }
      }
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int p = 0; p < 3; ++p) bp[cb][p] = b4[(it & 3) * 384 + (cb * 3 + p) * 64 + lane];
#define MF(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0)
    if (ORDER == 0) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        f32x16& c = acc[rb * CB + cb];
        MF(al[rb], bp[cb][0], c); MF(ah[rb], bp[cb][2], c); MF(am[rb], bp[cb][1], c);
        MF(am[rb], bp[cb][0], c); MF(ah[rb], bp[cb][1], c); MF(ah[rb], bp[cb][0], c);
      }
    } else {          // term-major: consecutive MFMAs hit different accumulators
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            f32x16& c = acc[rb * CB + cb];
            if (t == 0) MF(al[rb], bp[cb][0], c);
            if (t == 1) MF(ah[rb], bp[cb][2], c);
            if (t == 2) MF(am[rb], bp[cb][1], c);
            if (t == 3) MF(am[rb], bp[cb][0], c);
            if (t == 4) MF(ah[rb], bp[cb][1], c);
            if (t == 5) MF(ah[rb], bp[cb][0], c);
          }
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float sum = 0;
  for (int j = 0; j < RB * CB; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
  out[blockIdx.x * 256 + t] = sum;
  if (t == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

template <int RB, int CB, int SPLIT, int ORDER = 0>
void run(const float* in, float* out, unsigned long long* clk, int wps, int iters) {
  const int blocks = 256 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) k<RB, CB, SPLIT, ORDER><<<blocks, 256, 57344>>>(in, out, iters, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) k<RB, CB, SPLIT, ORDER><<<blocks, 256, 57344>>>(in, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2 * 1024];
  hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
  const double mf = (double)iters * RB * CB * 6;
  const double cyc_per = cyc / blocks / mf;
  printf("order=%d RB=%d CB=%d split=%d waves/SIMD=%d: %.1f core cycles per MFMA per wave -> pipe use %.0f%%; "
         "fp32-equivalent %.1f TFLOP/s by events; clock %.0f MHz\n",
         ORDER, RB, CB, SPLIT, wps, cyc_per, 100.0 * 32.0 * wps / cyc_per,
         (double)reps * blocks * 4 * mf / 6 * 32768.0 / ms / 1e9, cyc / wall * 100.0);
}
int main() {
  const int n = 1 << 20;
  float *in, *out; unsigned long long* clk;
  hipMalloc(&in, n * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&clk, 8 * 2048);
  float* h = (float*)malloc(n * 4);
  for (int i = 0; i < n; ++i) h[i] = rand() / (float)RAND_MAX * 2 - 1;
  hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
  for (int wps = 1; wps <= 2; ++wps) {
    run<1, 4, 1, 0>(in, out, clk, wps, 4000);
    run<1, 4, 1, 1>(in, out, clk, wps, 4000);
    run<1, 4, 0, 0>(in, out, clk, wps, 4000);
    run<1, 4, 0, 1>(in, out, clk, wps, 4000);
    run<1, 2, 1, 0>(in, out, clk, wps, 4000);
    run<1, 2, 1, 1>(in, out, clk, wps, 4000);
  }
  return 0;
}
