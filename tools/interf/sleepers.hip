// Occupancy-only co-runners for interference experiments (tools/bench_gemm_dw_mix.py):
// waves that sleep, spin on VALU, or stream memory for ~`us` microseconds.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void k_sleep(int ticks) {
  const unsigned long long w0 = wall_clock64();
  while (wall_clock64() - w0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(16);
}
extern "C" __global__ __launch_bounds__(256) void k_valu(int ticks, float* out) {
  const unsigned long long w0 = wall_clock64();
  float a = threadIdx.x, b = 1.0001f;
  while (wall_clock64() - w0 < (unsigned long long)ticks)
    for (int i = 0; i < 64; ++i) a = a * b + 0.5f;
  if (a == 12345.f) out[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void k_mem(int ticks, const float4* in, float4* out, int n) {
  const unsigned long long w0 = wall_clock64();
  int i = blockIdx.x * 256 + threadIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
  while (wall_clock64() - w0 < (unsigned long long)ticks) {
    const float4 v = in[i % n]; acc.x += v.x; acc.y += v.y; i += 256 * 1024;
  }
  if (acc.x == 12345.f) out[0] = acc;
}
extern "C" int launch_sleep(int blocks, int us, void* stream) {
  hipLaunchKernelGGL(k_sleep, dim3(blocks), dim3(256), 0, (hipStream_t)stream, us * 100);
  return (int)hipGetLastError();
}
extern "C" int launch_valu(int blocks, int us, float* out, void* stream) {
  hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, (hipStream_t)stream, us * 100, out);
  return (int)hipGetLastError();
}
extern "C" int launch_mem(int blocks, int us, const void* in, void* out, int n, void* stream) {
  hipLaunchKernelGGL(k_mem, dim3(blocks), dim3(256), 0, (hipStream_t)stream, us * 100,
                     (const float4*)in, (float4*)out, n);
  return (int)hipGetLastError();
}
