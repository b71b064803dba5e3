// Do CU-masked streams run concurrently with each other? 4 streams x 1 launch of 64 blocks spinning 500 us.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(int ticks, unsigned* sink) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && sink) sink[blockIdx.x] = 1;
}
static int run(const char* name, hipStream_t* s, int n, int blocks, int launches) {
  hipStream_t base; CK(hipStreamCreate(&base));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s[i], 100, nullptr);   // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, base));
  for (int i = 0; i < n; ++i) CK(hipStreamWaitEvent(s[i], e0, 0));
  for (int l = 0; l < launches; ++l)
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s[i], 50000 / launches, nullptr);
  for (int i = 0; i < n; ++i) { hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, s[i])); CK(hipStreamWaitEvent(base, ev, 0)); }
  CK(hipEventRecord(e1, base));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-52s %d streams x %d launches (500 us of spinning per stream): %.0f us\n", name, n, launches, ms * 1e3);
  return 0;
}
int main() {
  hipStream_t plain[8], masked[8], masked_same[4];
  for (int i = 0; i < 8; ++i) CK(hipStreamCreateWithFlags(&plain[i], hipStreamNonBlocking));
  for (int i = 0; i < 8; ++i) {
    uint32_t w[8] = {};
    for (int b = 32 * i; b < 32 * i + 32; ++b) w[b >> 5] |= 1u << (b & 31);
    CK(hipExtStreamCreateWithCUMask(&masked[i], 8, w));
  }
  for (int i = 0; i < 4; ++i) {
    uint32_t w[8] = {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0};
    CK(hipExtStreamCreateWithCUMask(&masked_same[i], 8, w));
  }
  for (int launches : {1, 10}) {
    if (run("plain streams", plain, 4, 64, launches)) return 1;
    if (run("masked streams, disjoint 32-CU partitions", masked, 4, 64, launches)) return 1;
    if (run("masked streams, disjoint 32-CU partitions", masked, 8, 64, launches)) return 1;
    if (run("masked streams, the SAME 64 CUs", masked_same, 4, 64, launches)) return 1;
    if (run("plain streams", plain, 8, 64, launches)) return 1;
  }
  return 0;
}
