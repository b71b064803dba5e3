// Prototype / accuracy probe: an fp32 x fp32 GEMM computed on the bf16 matrix pipe by
// splitting each fp32 operand EXACTLY into three bf16 pieces (8 + 8 + 8 significand
// bits) and summing 6 (or 9) piece products with v_mfma_f32_32x32x16_bf16, against the
// native v_mfma_f32_32x32x2_f32 and an fp64 host reference.
//   hipcc --offload-arch=gfx950 -O3 split_acc.hip -o split_acc
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const unsigned xb = __float_as_uint(x);
  h = xb >> 16;
  const float r1 = x - __uint_as_float(xb & 0xffff0000u);
  const unsigned r1b = __float_as_uint(r1);
  m = r1b >> 16;
  const float r2 = r1 - __uint_as_float(r1b & 0xffff0000u);
  l = __float_as_uint(r2) >> 16;
}

// one wave per 32x32 block of C; A [M,K] row-major, B [K,N] row-major
template <int TERMS, int ORDER>
__global__ __launch_bounds__(64) void gemm_split(const float* A, const float* B, float* C,
                                                 int M, int N, int K) {
  const int l = threadIdx.x, bm = blockIdx.y * 32, bn = blockIdx.x * 32;
  f32x16 acc, acc2;
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  for (int k0 = 0; k0 < K; k0 += 16) {
    u16x8 ah, am, al, bh, bm_, bl;
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + (l >> 5) * 8 + j;
      unsigned short h, m, lo;
      split3(k < K ? A[(size_t)(bm + (l & 31)) * K + k] : 0.f, h, m, lo);
      ah[j] = h; am[j] = m; al[j] = lo;
      split3(k < K ? B[(size_t)k * N + bn + (l & 31)] : 0.f, h, m, lo);
      bh[j] = h; bm_[j] = m; bl[j] = lo;
    }
#define MF(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0)
    if (ORDER == 0) {            // everything into one accumulator, small terms first
      if (TERMS >= 9) { MF(al, bl, acc); MF(am, bl, acc); MF(al, bm_, acc); }
      MF(al, bh, acc); MF(ah, bl, acc); MF(am, bm_, acc);
      MF(am, bh, acc); MF(ah, bm_, acc);
      MF(ah, bh, acc);
    } else {                     // corrections in their own accumulator, added at the end
      if (TERMS >= 9) { MF(al, bl, acc2); MF(am, bl, acc2); MF(al, bm_, acc2); }
      MF(al, bh, acc2); MF(ah, bl, acc2); MF(am, bm_, acc2);
      MF(am, bh, acc2); MF(ah, bm_, acc2);
      MF(ah, bh, acc);
    }
  }
  for (int r = 0; r < 16; ++r) {
    const int row = bm + (r >> 2) * 8 + (l >> 5) * 4 + (r & 3), col = bn + (l & 31);
    C[(size_t)row * N + col] = ORDER == 0 ? acc[r] : acc[r] + acc2[r];
  }
}

__global__ __launch_bounds__(64) void gemm_f32(const float* A, const float* B, float* C,
                                               int M, int N, int K) {
  const int l = threadIdx.x, bm = blockIdx.y * 32, bn = blockIdx.x * 32;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 2) {
    const int k = k0 + (l >> 5);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(bm + (l & 31)) * K + k],
                                               B[(size_t)k * N + bn + (l & 31)], acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = bm + (r >> 2) * 8 + (l >> 5) * 4 + (r & 3), col = bn + (l & 31);
    C[(size_t)row * N + col] = acc[r];
  }
}

static void report(const char* name, const std::vector<float>& c, const std::vector<double>& ref,
                   const std::vector<double>& mag) {
  double maxrel = 0, rms = 0, maxabs_scaled = 0;
  for (size_t i = 0; i < c.size(); ++i) {
    const double e = fabs(c[i] - ref[i]);
    maxrel = fmax(maxrel, e / fmax(fabs(ref[i]), 1e-30));
    maxabs_scaled = fmax(maxabs_scaled, e / mag[i]);      // relative to sum |a||b|
    rms += (e / mag[i]) * (e / mag[i]);
  }
  printf("  %-28s max |err|/sum|a||b| = %.3e   rms = %.3e   max rel = %.3e\n", name,
         maxabs_scaled, sqrt(rms / c.size()), maxrel);
}

int main() {
  const int shapes[3][3] = {{256, 256, 728}, {256, 128, 2048}, {256, 256, 256}};
  for (int dist = 0; dist < 2; ++dist)
    for (int s = 0; s < 3; ++s) {
      const int M = shapes[s][0], N = shapes[s][1], K = shapes[s][2];
      std::vector<float> a((size_t)M * K), b((size_t)K * N), c((size_t)M * N);
      srand(1 + s);
      for (auto& v : a) {
        const float u = rand() / (float)RAND_MAX;
        v = dist == 0 ? (u * 2 - 1) : fmaxf(0.f, u * 3 - 1);      // signed / ReLU-like
      }
      for (auto& v : b) v = (rand() / (float)RAND_MAX * 2 - 1) * 0.05f;
      std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
      for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
          double sacc = 0, m = 0;
          for (int k = 0; k < K; ++k) {
            const double p = (double)a[(size_t)i * K + k] * b[(size_t)k * N + j];
            sacc += p; m += fabs(p);
          }
          ref[(size_t)i * N + j] = sacc; mag[(size_t)i * N + j] = m;
        }
      float *dA, *dB, *dC;
      hipMalloc(&dA, a.size() * 4); hipMalloc(&dB, b.size() * 4); hipMalloc(&dC, c.size() * 4);
      hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice);
      hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice);
      printf("M=%d N=%d K=%d, A %s:\n", M, N, K, dist == 0 ? "uniform(-1,1)" : "relu-like");
      const dim3 g(N / 32, M / 32);
#define RUN(kern, name)                                                             \
  kern<<<g, 64>>>(dA, dB, dC, M, N, K);                                             \
  hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);                     \
  report(name, c, ref, mag);
      RUN(gemm_f32, "native fp32 MFMA 32x32x2");
      RUN((gemm_split<6, 0>), "split 6 terms, one acc");
      RUN((gemm_split<6, 1>), "split 6 terms, two acc");
      RUN((gemm_split<9, 0>), "split 9 terms, one acc");
      RUN((gemm_split<9, 1>), "split 9 terms, two acc");
      // host fp32 sequential fma chain for scale
      {
        for (int i = 0; i < M; ++i)
          for (int j = 0; j < N; ++j) {
            float sacc = 0;
            for (int k = 0; k < K; ++k) sacc = fmaf(a[(size_t)i * K + k], b[(size_t)k * N + j], sacc);
            c[(size_t)i * N + j] = sacc;
          }
        report("host fp32 fmaf chain", c, ref, mag);
      }
      hipFree(dA); hipFree(dB); hipFree(dC);
    }
  return 0;
}
