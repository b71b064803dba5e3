// How many independent VALU / LDS instructions hide behind one fp32 MFMA
// (v_mfma_f32_32x32x2_f32, 64 cycles) when ONE wave runs per SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NL>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
  __shared__ float lds[4096];
  const int t = threadIdx.x;
  lds[t] = in[t]; lds[t + 256] = in[t + 256];
  __syncthreads();
  float a = in[t & 1023], b = in[(t + 7) & 1023];
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = in[(t + i) & 1023];
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float4 l4 = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q & 7] = v[q & 7] * 1.0001f + 0.5f;   // VALU fma
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        float4 x = *reinterpret_cast<float4*>(&lds[((t * 4) + q * 256 + j * 64) & 4092 & ~3]);
        l4.x += x.x; l4.y += x.y;
      }
    }
  }
  float s = l4.x + l4.y;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + t] = s;
}
template <int NV, int NL> void run(const float* in, float* out, int wps = 1) {
  const int blocks = 256 * wps, iters = 20000;
  for (int r = 0; r < 3; ++r) k<NV, NL><<<blocks, 256>>>(in, out, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<NV, NL><<<blocks, 256>>>(in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = 5.0 * blocks * 4 * iters * 4 * 4096.0;
  printf("waves/SIMD=%d VALU/MFMA=%2d LDSb128/MFMA=%d : %.1f TFLOP/s\n", wps, NV, NL, flops / ms / 1e9);
}
int main() {
  float *in, *out; hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 22);
  hipMemset(in, 0, 1 << 20);
  run<0, 0>(in, out); run<2, 0>(in, out); run<4, 0>(in, out); run<8, 0>(in, out);
  run<12, 0>(in, out); run<16, 0>(in, out); run<24, 0>(in, out);
  run<0, 1>(in, out); run<0, 2>(in, out); run<4, 1>(in, out); run<8, 2>(in, out);
  run<0, 0>(in, out, 2); run<4, 0>(in, out, 2); run<8, 0>(in, out, 2); run<16, 0>(in, out, 2); run<0, 2>(in, out, 2); run<8, 2>(in, out, 2);
  run<8, 0>(in, out, 3); run<8, 2>(in, out, 3);
  return 0;
}
