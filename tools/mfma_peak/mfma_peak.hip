// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate on MI355X for different
// operand data (zeros vs random) and waves per SIMD. Build:
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xfffff]; b[i] = in[(t * 16 + 8 + i) & 0xfffff]; }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[(s + j) & 7], acc[j], 0, 0, 0);
  }
  float sum = 0;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) sum += acc[j][r];
  out[t] = sum;
}
int main() {
  const int n = 1 << 20;
  float *in, *out;
  hipMalloc(&in, n * 4); hipMalloc(&out, 256 * 8 * 1024 * 4);
  float* h = (float*)malloc(n * 4);
  for (int mode = 0; mode < 3; ++mode) {
    for (int i = 0; i < n; ++i) h[i] = mode == 0 ? 0.f : (mode == 1 ? (rand() / (float)RAND_MAX * 2 - 1) * 1e-3f : (rand() / (float)RAND_MAX * 2 - 1));
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 4; wps *= 2) {   // waves per SIMD = blocks per CU
      const int blocks = 256 * wps, iters = 4000;
      k<<<blocks, 256>>>(in, out, 10);
      hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      k<<<blocks, 256>>>(in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)blocks * 4 * iters * 32 * 4096.0;
      printf("data=%s waves/SIMD=%d  %.1f TFLOP/s  (%.2f ms)\n", mode == 0 ? "zeros" : mode == 1 ? "small" : "unit ", wps, flops / ms / 1e9, ms);
    }
  }
  // sustained: ~0.2 s back to back (power management settles), random operands
  for (int wps = 1; wps <= 4; wps += 1) {
    const int blocks = 256 * wps, iters = 4000, reps = 50 / wps + 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < reps; ++r) k<<<blocks, 256>>>(in, out, iters);   // warm
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<<<blocks, 256>>>(in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)reps * blocks * 4 * iters * 32 * 4096.0;
    printf("sustained data=unit waves/SIMD=%d  %.1f TFLOP/s over %.0f ms\n", wps, flops / ms / 1e9, ms);
  }
  return 0;
}
