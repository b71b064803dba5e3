// Micro-benchmark: v_mfma_f32_32x32x16_bf16 issue rate under the chip's power limit.
//   MODE 0: NACC independent accumulators cycled round robin, bare MFMAs
//   MODE 1: chains of 6 MFMAs on ONE accumulator (the split GEMM's old block order),
//           with FILL plain VALU ops pinned between consecutive MFMAs
//   MODE 2: the same 6 x 2 MFMAs alternating between TWO accumulators, same fillers
//   MODE 3: bare MFMAs (4 accumulators) + FILL ds_read_b128 per 24 MFMAs (the GEMM: 14)
//   MODE 4: bare MFMAs + FILL global_load_lds_dwordx4 (1 KB, L2-resident source) per 24
// The chip runs these loops at its 1400 W cap, so 1 / throughput is the ENERGY of an
// iteration: the price list of what rides along with the MFMAs.
// Reports TFLOP/s (bf16), core cycles per MFMA per wave and the core clock measured
// in-kernel (s_memtime per s_memrealtime tick). Build:
//   hipcc --offload-arch=gfx950 -O3 mfma_bf16.hip -o mfma_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC, int FILL>
__global__ __launch_bounds__(256) void k(const unsigned* in, float* out, int iters,
                                         unsigned long long* clk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = in[(t * 32 + i * 4 + j) & 0xfffff];
      b[i][j] = in[(t * 32 + 16 + i * 4 + j) & 0xfffff];
    }
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float f0 = __uint_as_float(a[0][0] & 0x3fffffffu), f1 = 1.0001f;
  __shared__ unsigned lds[16384];          // 64 KB
  u32x4 lacc = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = in[(blockIdx.x * 7 + i) & 0xfffff];
  __syncthreads();
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) {
#pragma unroll
      for (int s = 0; s < 24; ++s) {
        acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            __builtin_bit_cast(bf16x8, a[s & 3]), __builtin_bit_cast(bf16x8, b[(s >> 2) & 3]),
            acc[s & 3], 0, 0, 0);
        if (FILL > 0 && (s * FILL) / 24 != ((s + 1) * FILL) / 24) {
          __builtin_amdgcn_sched_barrier(0);
          if (MODE == 3) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(
                &lds[((it * 24 + s) & 15) * 256 + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 4096]);
            lacc[0] ^= r[0]; lacc[1] ^= r[1]; lacc[2] ^= r[2]; lacc[3] ^= r[3];
          } else {
            const unsigned dst = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
                (__attribute__((address_space(3))) unsigned*)lds)) +
                (((it * 24 + s) & 3) * 4 + (threadIdx.x >> 6)) * 1024;
            const unsigned* src = in + ((blockIdx.x * 977 + it * 131 + s * 17) & 0xfff) * 256 + (threadIdx.x & 63) * 4;
            if (MODE == 4) {
              asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                           : : "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory", "m0");
            } else {     // MODE 5: scalar base + 32-bit lane offset
              const unsigned* sb = in + ((blockIdx.x * 977 + it * 131 + s * 17) & 0xfff) * 256;
              const unsigned long long sbu = reinterpret_cast<unsigned long long>(sb);
              const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(sbu));
              const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(sbu >> 32));
              const unsigned long long su = (static_cast<unsigned long long>(hi) << 32) | lo;
              const unsigned voff = (threadIdx.x & 63) * 16;
              asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
                           : : "v"(voff), "s"(__builtin_amdgcn_readfirstlane(dst)), "s"(su) : "memory", "m0");
            }
            if ((s & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 24 / NACC; ++s)
#pragma unroll
        for (int j = 0; j < NACC; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, a[s & 3]), __builtin_bit_cast(bf16x8, b[(s + j) & 3]),
              acc[j], 0, 0, 0);
    } else {
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int s = 0; s < 12; ++s) {
          const int j = MODE == 1 ? 2 * g + (s / 6) : 2 * g + (s & 1);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, a[s & 3]), __builtin_bit_cast(bf16x8, b[(s + g) & 3]),
              acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f0) : "v"(f1));
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = f0 + __uint_as_float((lacc[0] ^ lacc[1] ^ lacc[2] ^ lacc[3]) & 0x3fffffffu) + __uint_as_float(lds[threadIdx.x] & 0x3fffffffu);
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
  out[t] = sum;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <int MODE, int NACC, int FILL>
void run(const unsigned* in, float* out, unsigned long long* clk, int wps, int iters, int reps) {
  const int blocks = 256 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) k<MODE, NACC, FILL><<<blocks, 256>>>(in, out, iters, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<MODE, NACC, FILL><<<blocks, 256>>>(in, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[2 * 1024];
  hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
  const double mfmas = (double)iters * 24;
  printf("mode=%d acc=%d fill=%2d waves/SIMD=%d: %7.1f TFLOP/s bf16 (%5.1f fp32-equivalent /6) | %.1f core cycles/MFMA/wave, core clock %.0f MHz\n",
         MODE, NACC, FILL, wps, (double)reps * blocks * 4 * mfmas * 32768.0 / ms / 1e9,
         (double)reps * blocks * 4 * mfmas * 32768.0 / ms / 1e9 / 6.0,
         cyc / blocks / mfmas, cyc / wall * 100.0);
}
int main() {
  const int n = 1 << 20;
  unsigned* in; float* out; unsigned long long* clk;
  hipMalloc(&in, n * 4); hipMalloc(&out, 256 * 8 * 1024 * 4); hipMalloc(&clk, 8 * 2048);
  unsigned* h = (unsigned*)malloc(n * 4);
  for (int i = 0; i < n; ++i) {       // random bf16 pairs in (-2, 2)
    const unsigned x = 0x3f80u | (rand() & 0x807f), y = 0x3f80u | (rand() & 0x807f);
    h[i] = x | (y << 16);
  }
  hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
  for (int wps = 1; wps <= 2; ++wps) {
    run<0, 4, 0>(in, out, clk, wps, 20000, 5);
    run<0, 8, 0>(in, out, clk, wps, 20000, 5);
    run<1, 4, 0>(in, out, clk, wps, 20000, 5);
    run<1, 4, 1>(in, out, clk, wps, 20000, 5);
    run<1, 4, 3>(in, out, clk, wps, 20000, 5);
    run<1, 4, 6>(in, out, clk, wps, 20000, 5);
    run<3, 4, 6>(in, out, clk, wps, 20000, 5);
    run<3, 4, 14>(in, out, clk, wps, 20000, 5);
    run<3, 4, 24>(in, out, clk, wps, 20000, 5);
    run<4, 4, 3>(in, out, clk, wps, 20000, 5);
    run<4, 4, 5>(in, out, clk, wps, 20000, 5);
    run<4, 4, 8>(in, out, clk, wps, 20000, 5);
    run<5, 4, 3>(in, out, clk, wps, 20000, 5);
    run<5, 4, 5>(in, out, clk, wps, 20000, 5);
    run<5, 4, 8>(in, out, clk, wps, 20000, 5);
    run<2, 4, 0>(in, out, clk, wps, 20000, 5);
    run<2, 4, 1>(in, out, clk, wps, 20000, 5);
    run<2, 4, 3>(in, out, clk, wps, 20000, 5);
    run<2, 4, 6>(in, out, clk, wps, 20000, 5);
  }
  return 0;
}
