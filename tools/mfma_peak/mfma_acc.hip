// Micro-benchmark: v_mfma_f32_32x32x2_f32 issue rate as a function of the number of
// independent accumulators a wave cycles through (1, 2, 4), waves per SIMD and
// launch length, with the shader clock measured in-kernel (s_memtime ticks per
// s_memrealtime tick x 100 MHz). Build:
//   hipcc --offload-arch=gfx950 -O3 mfma_acc.hip -o mfma_acc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters,
                                         unsigned long long* clk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xfffff]; b[i] = in[(t * 16 + 8 + i) & 0xfffff]; }
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 32 / NACC; ++s)
#pragma unroll
      for (int j = 0; j < NACC; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 7], b[(s + j) & 7], acc[j], 0, 0, 0);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float sum = 0;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
  out[t] = sum;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <int NACC>
void run(const float* in, float* out, unsigned long long* clk, int wps, int iters, int reps) {
  const int blocks = 256 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<NACC><<<blocks, 256>>>(in, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2 * 1024];
  hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
  const double mfmas = (double)iters * 32;
  printf("acc=%d waves/SIMD=%d iters=%5d reps=%3d: %6.1f TFLOP/s by events | in-kernel: %.1f core cycles/MFMA/wave, %.2f ns/MFMA/wave, core clock %.0f MHz\n",
         NACC, wps, iters, reps, (double)reps * blocks * 4 * mfmas * 4096.0 / ms / 1e9,
         cyc / blocks / mfmas, wall / blocks * 10.0 / mfmas, cyc / wall * 100.0);
}
int main() {
  const int n = 1 << 20;
  float *in, *out; unsigned long long* clk;
  hipMalloc(&in, n * 4); hipMalloc(&out, 256 * 8 * 1024 * 4); hipMalloc(&clk, 8 * 2048);
  float* h = (float*)malloc(n * 4);
  for (int i = 0; i < n; ++i) h[i] = rand() / (float)RAND_MAX * 2 - 1;
  hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
  for (int pass = 0; pass < 2; ++pass)
    for (int wps = 1; wps <= 2; ++wps) {
      // short launches (~25 us of MFMAs, like one GEMM workgroup) and long ones
      run<1>(in, out, clk, wps, 23, 1);  run<2>(in, out, clk, wps, 23, 1);  run<4>(in, out, clk, wps, 23, 1);
      run<1>(in, out, clk, wps, 23, 40); run<2>(in, out, clk, wps, 23, 40); run<4>(in, out, clk, wps, 23, 40);
      run<1>(in, out, clk, wps, 4000, 5); run<2>(in, out, clk, wps, 4000, 5); run<4>(in, out, clk, wps, 4000, 5);
    }
  return 0;
}
