// Shared pieces of the pointwise-GEMM kernels (pointwise_gemm.hip: the C entry points and the
// routing; pointwise_gemm_h2.hip / _split.hip: the product kernels; ref/*.hip: the fp32-MFMA
// reference kernels of the test build).
// Everything here has internal linkage: each translation unit gets its own copy.
#pragma once
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"

namespace epos {

// The fp32-MFMA kernels (v_mfma_f32_32x32x2_f32; rounds 1-2) are TEST-ONLY since round 6:
// ref/pointwise_gemm_dma.hip (LDS-DMA data path) and ref/pointwise_gemm_staged.hip
// (register-staged, pre-activation ReLU) are linked into libepos_hip_ref.so only and announce
// themselves here when they are (static initialisers); the product library leaves both null
// and refuses a problem without Wh / Ws. Same argument meaning as launch_grouped_h2 below:
// conv_cin != nullptr: implicit 3x3 conv, problem i has conv_cin[i] input channels and
// dilation conv_rate[i] (K = 9 * conv_cin[i], A = the NHWC input of Hi x Wi pixels,
// `sub` = stride, M = B * Ho * Wo output pixels).
struct Fp32MfmaRef {
  int (*dma)(const EposPointwiseArgs* args, int count, hipStream_t s, const int* conv_cin,
             const int* conv_rate);
  int (*staged)(const EposPointwiseArgs* args, int count, hipStream_t s);
};
Fp32MfmaRef& fp32_mfma_ref();        // pointwise_gemm.hip
// pointwise_gemm_split.hip: fp32 GEMM on the bf16 matrix pipe (exact three-way operand
// split, six piece products); every problem carries split-packed weights (p.Ws).
int launch_grouped_split(const EposPointwiseArgs* args, int count, hipStream_t s,
                         const int* conv_cin = nullptr, const int* conv_rate = nullptr);
bool split_eligible(const EposPointwiseArgs* args, int count);
// pointwise_gemm_h2.hip: fp32 GEMM on the fp16 matrix pipe (two fp16 pieces per operand,
// three piece products); every problem carries fp16-pair weights (p.Wh). A problem
// without an absmax slot (p.a_amax) gets one from the library's ring, measured first.
int launch_grouped_h2(const EposPointwiseArgs* args, int count, hipStream_t s,
                      const int* conv_cin = nullptr, const int* conv_rate = nullptr);
bool h2_eligible(const EposPointwiseArgs* args, int count);

namespace {

// More than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel:
// once per DEVICE and kernel instantiation (the caller's function-local statics), under a lock
// -- launches may come from several host threads and devices.
struct LdsAttrOnce {
  std::mutex mu;
  bool set[16] = {};
};
inline int ensure_dynamic_lds(LdsAttrOnce& once, const void* kern, int bytes, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
    set_error("%s: no current device", what);
    return EPOS_E_INVALID;
  }
  std::lock_guard<std::mutex> lock(once.mu);
  if (once.set[dev]) return EPOS_OK;
  const int rc = check_hip(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               bytes), what);
  if (!rc) once.set[dev] = true;
  return rc;
}

constexpr int BN = 128;
constexpr int BK = 32;
constexpr int LDS_A_ROW = BK + 4;                  // floats
constexpr int LDS_B_TILE = (BK / 4) * BN * 4;      // floats per buffer
constexpr int THREADS = 256;
constexpr int MAX_GROUP = 8;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GroupedArgs {
  EposPointwiseArgs p[MAX_GROUP];
  int tile_start[MAX_GROUP + 1];   // prefix sum of tiles per problem
  int tiles_n[MAX_GROUP];
  int npad[MAX_GROUP];
  int conv_cin[MAX_GROUP];         // LDS-DMA kernel, implicit 3x3 conv: input channels
  int conv_rate[MAX_GROUP];        //   and dilation
  int count;
  // fp16-pair kernel: n / tiles_n[i] by multiply-shift (set by launch_grouped_h2; a run-time
  // division costs ~25 instructions of every workgroup's set-up)
  unsigned tn_mul[MAX_GROUP], tn_sh1[MAX_GROUP], tn_sh2[MAX_GROUP];
  // 16 bytes of zeros in device memory (source of the LDS-DMA pieces that lie outside the
  // matrix); as a kernel argument it costs no extra scalar-load round trip, as a __device__
  // variable its address comes through the GOT: two more dependent loads in every
  // workgroup's set-up
  const float* zero_chunk;
};

// ---- LDS-DMA (global_load_lds_dwordx4) helpers, inline asm: hipcc neither waits for
// these before ds_reads nor counts them, the kernels' vmcnt waits are explicit.
// one wave-instruction: 64 lanes x 16 B -> LDS [lds_dst, lds_dst + 1024)
__device__ __forceinline__ void glds16_v(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const uint64_t b = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(b >> 32));
  return reinterpret_cast<const float*>((static_cast<uint64_t>(hi) << 32) | lo);
}
__device__ __forceinline__ void glds16_s(unsigned voff, const float* sbase,
                                         unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}

// The same without saving M0 (nothing else in these kernels reads it).
__device__ __forceinline__ void glds16_v_m0(const float* gsrc, unsigned lds_dst) {
#ifdef EPOS_SPLIT_ABL_SAMEM0      // ablation: no M0 write (every piece lands on one spot)
  asm volatile("global_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
#else
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
#endif
}
__device__ __forceinline__ void glds16_s_m0(unsigned voff, const float* sbase,
                                            unsigned lds_dst) {
#ifdef EPOS_SPLIT_ABL_SAMEM0
  asm volatile("global_load_lds_dwordx4 %0, %2"
               : : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory", "m0");
#else
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
               : : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory", "m0");
#endif
}
// LDS-DMA with an instruction offset: the immediate is added to the global address AND to
// the LDS address (LDS_ADDR = M0 + inst_offset + lane * 16), so pieces that are 1 KB apart
// in LDS share one M0 write when the source pointer is moved back by the same amount.
template <int OFF>
__device__ __forceinline__ void glds16_v_off(const float* gsrc) {
  asm volatile("global_load_lds_dwordx4 %0, off offset:%1" : : "v"(gsrc), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void glds16_s_off(unsigned voff, const float* sbase) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2"
               : : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// max over the wave of a non-negative value -> one atomic max on word (salt % 64) of an
// absmax slot (include/epos_hip.h). NaNs are ignored by fmaxf; +Inf propagates.
__device__ __forceinline__ void amax_publish(unsigned* slot, float m, int lane, int salt) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0)
    __hip_atomic_fetch_max(slot + (salt & (EPOS_AMAX_WORDS - 1)), __float_as_uint(m),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f),
                     fmaxf(v.w, 0.f));
}

// ---------------------------------------------------------------------------
// Vectorised epilogue. The MFMA accumulator layout gives a lane ONE output column
// and 16 rows, i.e. 4-byte stores that touch 128 B per row -- store-ISSUE bound
// (a 14 MB layer output took ~9 us). The wave instead transposes its tile through
// its own LDS region (bias added on the way in) and then streams it out row-major
// as float4: 16 B per lane, the residual is fetched the same way, ReLU last.
// Falls back to the scalar path when rows are not 16-byte aligned (e.g. the
// 22-channel object head).
// ---------------------------------------------------------------------------
constexpr int EP_ROW = 68;     // floats per staged row (64 + 4: 16 B aligned, no conflicts)

__device__ __forceinline__ bool vec_epilogue_ok(const EposPointwiseArgs& p, bool has_res) {
  bool ok = (p.ldc & 3) == 0 && (p.N & 3) == 0 &&
            (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;
  if (has_res)
    ok = ok && (p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(p.R) & 15) == 0;
  return ok;
}

// acc: TM x TN accumulator tiles of this wave (row-tile major); the wave tile is
// (TM*32) rows x (TN*32) columns at (m0w, n0w); `ws` = TM*32*EPR floats of LDS
// owned by this wave (EPR = TN*32 + 4 floats per staged row).
template <int TM, int TN, bool HAS_RES, int EPR = EP_ROW>
__device__ __forceinline__ void vec_epilogue(float* ws, const f32x16* acc,
                                             const EposPointwiseArgs& p, int m0w,
                                             int n0w, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
  const int M = p.M, N = p.N;
  constexpr int C4 = TN * 8;                 // float4 per staged row
  constexpr int RPI = 64 / C4;               // rows per wave instruction
  constexpr int NI = TM * 32 / RPI;
  const int c4 = lane % C4, r0 = lane / C4;
  const int n = n0w + c4 * 4;
  // The residual rows are requested first, unconditionally and from clamped
  // addresses, so that all of them are in flight together (a load under the
  // store predicate would be waited for on the spot, one row at a time).
  float4 rv[HAS_RES ? NI : 1];
  if (HAS_RES) {
    const int ncl = n < N ? n : 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int m = m0w + r0 + i * RPI;
      m = m < M ? m : M - 1;
      rv[i] = *reinterpret_cast<const float4*>(p.R + static_cast<int64_t>(m) * p.ldr + ncl);
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nb = n0w + j * 32 + l31;
    const float bias = p.bias ? p.bias[nb < N ? nb : N - 1] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ws[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * EPR + j * 32 + l31] =
            acc[i * TN + j][r] + bias;
  }
  // same wave wrote and reads: only the LDS counter has to drain
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  const bool relu = p.relu != 0;
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = r0 + i * RPI;
    const int m = m0w + row;
    float4 v = *reinterpret_cast<const float4*>(ws + row * EPR + c4 * 4);
    if (HAS_RES) {
      v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;
    }
    if (relu) v = relu4(v);
    if (m < M && n < N) {
      *reinterpret_cast<float4*>(p.C + static_cast<int64_t>(m) * p.ldc + n) = v;
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  }
  // absmax slot of the output (EposPointwiseArgs.c_amax): the consumers' fp16-pair scale
  if (p.c_amax) amax_publish(p.c_amax, amax, lane, static_cast<int>(blockIdx.x) * 4 + (m0w >> 5));
}

}  // namespace
}  // namespace epos
