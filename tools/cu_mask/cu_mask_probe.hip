// Which CUs does a CU-masked HIP stream run on (gfx950, 8 XCDs x 32 CUs)? Probes the bit -> (XCD, SE, CU)
// layout of hipExtStreamCreateWithCUMask and whether a hipGraph launched INTO a masked stream honours it.
//   hipcc --offload-arch=gfx950 -O2 -o tools/cu_mask/cu_mask_probe tools/cu_mask/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(unsigned* out, int spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(4);   // hold the slot
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 15u; out[2 * blockIdx.x + 1] = hw; }
}

static int report(const char* name, hipStream_t s, unsigned* dbuf, int nblk, bool via_graph) {
  std::vector<unsigned> h(2 * nblk);
  if (via_graph) {
    hipStream_t cap; CK(hipStreamCreate(&cap));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, cap, dbuf, 2000);
    CK(hipStreamEndCapture(cap, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(cap));
  } else {
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(256), 0, s, dbuf, 2000);
    CK(hipStreamSynchronize(s));
  }
  CK(hipMemcpy(h.data(), dbuf, h.size() * 4, hipMemcpyDeviceToHost));
  std::set<unsigned> cus; int per_xcc[16] = {};
  for (int b = 0; b < nblk; ++b) {
    const unsigned xcc = h[2 * b], hw = h[2 * b + 1];
    const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
    per_xcc[xcc & 15]++;
  }
  printf("%-44s %s: %3zu distinct (xcc,se,sh,cu); blocks per XCC:", name, via_graph ? "graph " : "direct", cus.size());
  for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
  printf("\n");
  return 0;
}

int main() {
  const int nblk = 2048;
  unsigned* dbuf; CK(hipMalloc(&dbuf, 2 * nblk * 4));
  hipStream_t plain; CK(hipStreamCreate(&plain));
  if (report("no mask", plain, dbuf, nblk, false)) return 1;
  struct M { const char* name; uint32_t w[8]; };
  std::vector<M> masks;
  { M m = {"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}}; masks.push_back(m); }
  { M m = {"bits 0..63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0}}; masks.push_back(m); }
  { M m = {"bits 32..63", {0, 0xffffffffu, 0, 0, 0, 0, 0, 0}}; masks.push_back(m); }
  { M m = {"bits 224..255", {0, 0, 0, 0, 0, 0, 0, 0xffffffffu}}; masks.push_back(m); }
  { M m = {"bits = 0 mod 8", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}}; masks.push_back(m); }
  { M m = {"bits = 0,1 mod 8", {0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u}}; masks.push_back(m); }
  { M m = {"bits = 0 mod 4", {0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u}}; masks.push_back(m); }
  { M m = {"bits 0..127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}}; masks.push_back(m); }
  { M m = {"bit 0 only", {1u, 0, 0, 0, 0, 0, 0, 0}}; masks.push_back(m); }
  { M m = {"bit 8 only", {0x100u, 0, 0, 0, 0, 0, 0, 0}}; masks.push_back(m); }
  for (auto& m : masks) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.w);
    if (e != hipSuccess) { printf("%-44s create failed: %s\n", m.name, hipGetErrorString(e)); continue; }
    if (report(m.name, s, dbuf, nblk, false)) return 1;
    if (report(m.name, s, dbuf, nblk, true)) return 1;
    CK(hipStreamDestroy(s));
  }
  // four masked streams at once: do they run concurrently on disjoint CUs?
  {
    hipStream_t s[4];
    for (int i = 0; i < 4; ++i) {
      uint32_t w[8];
      for (int j = 0; j < 8; ++j) w[j] = 0x03030303u << (2 * i);     // bits = 2i, 2i+1 mod 8
      CK(hipExtStreamCreateWithCUMask(&s[i], 8, w));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned* d4; CK(hipMalloc(&d4, 4 * 2 * 512 * 4));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, plain));
    for (int i = 0; i < 4; ++i) CK(hipStreamWaitEvent(s[i], e0, 0));
    for (int rep = 0; rep < 20; ++rep)
      for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(probe, dim3(512), dim3(256), 0, s[i], d4 + i * 1024, 1000);   // 10 us spin
    for (int i = 0; i < 4; ++i) { hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, s[i])); CK(hipStreamWaitEvent(plain, ev, 0)); }
    CK(hipEventRecord(e1, plain));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("4 masked streams x 20 launches of 512 blocks spinning 10 us (64 CUs x 8 blocks/CU each): %.1f us total\n", ms * 1e3);
  }
  return 0;
}
