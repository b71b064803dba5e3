// Microbenchmark: how fast can ONE CU pull the depthwise kernels' access pattern -- wave
// instructions of 64 lanes x 16 B that cover 16 pixels x 64 B (4 lanes per pixel, pixel
// rows `ld` floats apart) or 4 pixels x 256 B, or 1 KB contiguous -- out of L2 / Infinity
// Cache, (a) as ordinary global loads into VGPRs (what depthwise3x3_s1_kernel issues) and
// (b) as LDS-DMA (global_load_lds_dwordx4), at 4 / 8 / 16 waves per CU? The answer bounds
// any depthwise design (stand-alone or fused): bytes per output x this rate.
//   hipcc --offload-arch=gfx950 -O3 load_rate.hip -o load_rate && ./load_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void dma16(const float* src, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               : : "v"(src), "s"(lds_dst) : "memory", "m0");
}

// LPP: lanes per pixel (4 -> 64 B per pixel, 16 -> 256 B, 64 -> 1 KB contiguous)
template <int MODE, int LPP>
__global__ __launch_bounds__(256) void k(const float* x, int ld, int npix, int iters,
                                         float* sink) {
  extern __shared__ float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) float*)smem)) + wave * 8192;
  const int ppi = 64 / LPP;                       // pixels per wave instruction
  // this CU's private pixel range (so that different workgroups do not share lines)
  const int chunk = npix / gridDim.x;
  const int p0 = blockIdx.x * chunk;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int pos = wave * ppi * 8;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int px = p0 + (pos + u * ppi + lane / LPP) % chunk;
      const float* src = x + static_cast<size_t>(px) * ld + (lane % LPP) * 4 + ((it & 3) * LPP * 4) % 512;
      if (MODE == 0) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      } else {
        dma16(src, __builtin_amdgcn_readfirstlane(lds0 + u * 1024));
      }
    }
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    pos += 4 * ppi * 8;
  }
  if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc.x == 123.f) sink[t] = acc.y + acc.z + acc.w + smem[t];
}

template <int MODE, int LPP>
void run(const float* x, int ld, int npix, float* sink, int wgs_per_cu, const char* name) {
  const int cus = 256, iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds = wgs_per_cu == 1 ? 65536 : wgs_per_cu == 2 ? 65536 : 32768;   // occupancy
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, LPP>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, LPP>), dim3(cus * wgs_per_cu), dim3(256), lds, 0, x, ld, npix,
                       iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = 1.0 * cus * wgs_per_cu * 4 * iters * 8;
  const double ns_per_instr_per_cu = ms * 1e6 / (wave_instr / cus);
  printf("%-34s %2d waves/CU: %6.1f ns per 1 KB wave instruction per CU, %6.2f TB/s chip\n", name,
         4 * wgs_per_cu, ns_per_instr_per_cu, wave_instr * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
  const int ld = 728, npix = 4800 * 4;            // 56 MB: Infinity-Cache resident
  float *x, *sink;
  hipMalloc(&x, sizeof(float) * static_cast<size_t>(npix) * ld + 4096);
  hipMalloc(&sink, 4096);
  hipMemset(x, 0, sizeof(float) * static_cast<size_t>(npix) * ld + 4096);
  for (int w = 1; w <= 4; w *= 2) {
    run<0, 4>(x, ld, npix, sink, w, "VGPR loads, 16 px x 64 B");
    run<1, 4>(x, ld, npix, sink, w, "LDS-DMA,    16 px x 64 B");
    run<0, 16>(x, ld, npix, sink, w, "VGPR loads, 4 px x 256 B");
    run<1, 16>(x, ld, npix, sink, w, "LDS-DMA,    4 px x 256 B");
    run<0, 64>(x, ld, npix, sink, w, "VGPR loads, 1 KB contiguous");
    run<1, 64>(x, ld, npix, sink, w, "LDS-DMA,    1 KB contiguous");
  }
  return 0;
}
