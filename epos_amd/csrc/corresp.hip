// Many-to-many 2D-3D correspondence extraction on gfx950 -- replaces
// epos_lib/corresp.py:9-101 (establish_many_to_many) and misc.py:14-26.
//
// The reference masks pixels by object confidence, keeps for every masked pixel
// the fragments with conf > max_conf * tau_b and emits one correspondence per
// (pixel, kept fragment) in raster order, then ascending fragment id
// (np.nonzero order, corresp.py:52,67). Here that is a stable stream compaction
// in three launches, all integer-exact:
//   1. corr_mask:  one wave per 64 pixels; lane = pixel for the object-confidence
//                  test (ballot), then lane = fragment (F <= 64) for each masked
//                  pixel: wave max, threshold, ballot -> 64-bit kept-fragment mask.
//   2. corr_scan:  per slot exclusive scans (masked-pixel index, first row).
//   3. corr_fill:  lane = fragment; row = slot base + pixel base + rank of the
//                  fragment inside the mask (popcount of lower bits).
// One "slot" = one (image, object) pair; all slots of a batch go in one launch.
// HBM-bound byte/compare work: no MFMA here.
#include "common.h"

namespace epos {
namespace {

__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Pixels per wave in the mask / fill kernels: the per-pixel part uses the first
// CORR_PPW lanes, the per-fragment part all 64. A wave walks its masked pixels one after
// the other (each step a dependent 256-byte load), so FEWER pixels per wave = more
// waves in flight = the latency of those steps overlaps.
constexpr int CORR_PPW = 16;

__global__ __launch_bounds__(256) void corr_mask_kernel(
    const float* __restrict__ obj_confs, const float* __restrict__ frag_confs,
    const EposCorrSlot* __restrict__ slots, int P, int O, int F, float tau_a,
    float tau_b, int32_t* px_flag, int32_t* corr_cnt, uint64_t* frag_mask) {
  const int s = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int p0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * CORR_PPW;
  if (p0 >= P) return;                       // wave-uniform
  const int img = slots[s].image, obj = slots[s].obj_id;
  const int p = p0 + lane;
  const bool mine = lane < CORR_PPW && p < P;   // this lane owns a pixel
  const int64_t pix0 = static_cast<int64_t>(img) * P;
  bool masked = false;
  if (mine) masked = obj_confs[(pix0 + p) * (O + 1) + obj] > tau_a;  // corresp.py:46-47
  uint64_t todo = __ballot(masked);
  uint64_t mybits = 0;
  while (todo) {                             // wave-uniform loop over masked pixels
    const int j = __ffsll(static_cast<long long>(todo)) - 1;
    todo &= todo - 1;
    const float* fc = frag_confs + ((pix0 + p0 + j) * O + (obj - 1)) * F;
    const float v = lane < F ? fc[lane] : -INFINITY;
    const float m = wave_max64(v);           // corresp.py:63
    const float thr = m * tau_b;             // f32 * f32 (numpy weak-scalar rule)
    const uint64_t bits = __ballot(lane < F && v > thr);   // corresp.py:64, strict >
    if (lane == j) mybits = bits;
  }
  if (mine) {
    const int64_t o = static_cast<int64_t>(s) * P + p;
    px_flag[o] = masked ? 1 : 0;
    corr_cnt[o] = __popcll(mybits);
    frag_mask[o] = mybits;
  }
}

// In-place exclusive scan of two int arrays of length P per slot: one 1024-thread
// workgroup per slot walks the arrays in coalesced 1024-element chunks; inside a chunk
// a wave-level inclusive scan (shuffle up) + the 16 wave totals through LDS; the running
// carry is a register. (The first version gave every thread a private strided
// sub-range and scanned the 1024 partial sums with 20 barriers: 41 us.)
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(v, o, 64);
    if (lane >= o) v += u;
  }
  return v;
}

template <int VEC>
__global__ __launch_bounds__(1024) void corr_scan_kernel(int32_t* px, int32_t* cnt,
                                                         int P, int32_t* totals) {
  __shared__ int32_t wa[16], wb[16];
  const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int32_t* a = px + static_cast<int64_t>(s) * P;
  int32_t* b = cnt + static_cast<int64_t>(s) * P;
  int32_t carry_a = 0, carry_b = 0;
  for (int base = 0; base < P; base += 1024 * VEC) {
    const int i = base + t * VEC;                // VEC consecutive elements per thread
    int32_t va[VEC], vb[VEC];
    if (VEC == 4 && i < P) {                     // P % 4 == 0: all four are in range
      const int4 xa = *reinterpret_cast<const int4*>(a + i);
      const int4 xb = *reinterpret_cast<const int4*>(b + i);
      va[0] = xa.x; va[VEC > 1 ? 1 : 0] = xa.y; va[VEC > 2 ? 2 : 0] = xa.z; va[VEC > 3 ? 3 : 0] = xa.w;
      vb[0] = xb.x; vb[VEC > 1 ? 1 : 0] = xb.y; vb[VEC > 2 ? 2 : 0] = xb.z; vb[VEC > 3 ? 3 : 0] = xb.w;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        va[e] = i + e < P ? a[i + e] : 0;
        vb[e] = i + e < P ? b[i + e] : 0;
      }
    }
    int32_t ta = 0, tb = 0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { ta += va[e]; tb += vb[e]; }
    const int32_t ia = wave_incl_scan(ta, lane), ib = wave_incl_scan(tb, lane);
    if (lane == 63) { wa[wave] = ia; wb[wave] = ib; }
    __syncthreads();
    int32_t off_a = carry_a, off_b = carry_b, tot_a = 0, tot_b = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int32_t xa = wa[w], xb = wb[w];
      if (w < wave) { off_a += xa; off_b += xb; }
      tot_a += xa; tot_b += xb;
    }
    int32_t ra = off_a + ia - ta, rb = off_b + ib - tb;     // exclusive prefix of the thread
    int32_t oa[VEC], ob[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { oa[e] = ra; ob[e] = rb; ra += va[e]; rb += vb[e]; }
    if (VEC == 4 && i < P) {
      *reinterpret_cast<int4*>(a + i) = make_int4(oa[0], oa[VEC > 1 ? 1 : 0], oa[VEC > 2 ? 2 : 0], oa[VEC > 3 ? 3 : 0]);
      *reinterpret_cast<int4*>(b + i) = make_int4(ob[0], ob[VEC > 1 ? 1 : 0], ob[VEC > 2 ? 2 : 0], ob[VEC > 3 ? 3 : 0]);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (i + e < P) { a[i + e] = oa[e]; b[i + e] = ob[e]; }
    }
    carry_a += tot_a; carry_b += tot_b;
    __syncthreads();
  }
  if (t == 0) { totals[2 * s] = carry_a; totals[2 * s + 1] = carry_b; }
}

__global__ void corr_slot_bases_kernel(const int32_t* totals, int S,
                                       int64_t* slot_base) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int64_t acc = 0;
    for (int s = 0; s < S; ++s) { slot_base[s] = acc; acc += totals[2 * s + 1]; }
    slot_base[S] = acc;
  }
}

__global__ __launch_bounds__(256) void corr_fill_kernel(
    const float* __restrict__ obj_confs, const float* __restrict__ frag_confs,
    const float* __restrict__ frag_coords, const double* __restrict__ centers,
    const double* __restrict__ sizes, const EposCorrSlot* __restrict__ slots,
    int P, int W, int O, int F, double inv_scale, const int32_t* px_off,
    const int32_t* corr_off, const uint64_t* frag_mask, const int64_t* slot_base,
    int64_t capacity, EposCorrOut out, int32_t* overflow) {
  const int s = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int p0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * CORR_PPW;
  if (p0 >= P) return;
  const int img = slots[s].image, obj = slots[s].obj_id;
  const int p = p0 + lane;
  const int64_t pix0 = static_cast<int64_t>(img) * P;
  uint64_t mbits = 0;
  int32_t pxo = 0, co = 0;
  if (lane < CORR_PPW && p < P) {
    const int64_t o = static_cast<int64_t>(s) * P + p;
    mbits = frag_mask[o]; pxo = px_off[o]; co = corr_off[o];
  }
  uint64_t todo = __ballot(mbits != 0);
  const int64_t base_s = slot_base[s];
  const double* cen = centers + static_cast<int64_t>(obj - 1) * F * 3;
  const double* siz = sizes + static_cast<int64_t>(obj - 1) * F;
  while (todo) {
    const int j = __ffsll(static_cast<long long>(todo)) - 1;
    todo &= todo - 1;
    const uint64_t bits = __shfl(mbits, j, 64);
    const int32_t px_id = __shfl(pxo, j, 64);
    const int32_t row0 = __shfl(co, j, 64);
    if ((bits >> lane) & 1ull) {
      const int rank = __popcll(bits & ((1ull << lane) - 1ull));
      const int64_t row = base_s + row0 + rank;
      if (row >= capacity) {
        *overflow = 1;
      } else {
      const int pj = p0 + j;
      const int y = pj / W, x = pj - y * W;
      const int64_t fidx = ((pix0 + pj) * O + (obj - 1)) * F + lane;
      const float conf_obj = obj_confs[(pix0 + pj) * (O + 1) + obj];
      const float conf_frag = frag_confs[fidx];
      out.px_id[row] = px_id;
      out.frag_id[row] = lane;
      // misc.py:26: scale * (idx + 0.5), x first (corresp.py:55-57).
      out.coord_2d[2 * row + 0] = inv_scale * (static_cast<double>(x) + 0.5);
      out.coord_2d[2 * row + 1] = inv_scale * (static_cast<double>(y) + 0.5);
      const double sz = siz[lane];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        // corresp.py:76-78: the in-place f32 `*=` rounds the f64 product to f32
        // before the f64 add.
        const float local = static_cast<float>(
            static_cast<double>(frag_coords[fidx * 3 + d]) * sz);
        out.coord_3d[3 * row + d] = cen[lane * 3 + d] + static_cast<double>(local);
      }
      out.conf_obj[row] = conf_obj;
      out.conf_frag[row] = conf_frag;
      out.conf[row] = conf_obj * conf_frag;                  // corresp.py:82-84
      }
    }
  }
}

// --------------------------------------------------------------------------
// project_to_surface (corresp.py:87-88 -> datagen.py:128-154): closest point of the
// object's triangle mesh for every predicted 3D point. The reference asks libigl's AABB
// tree; here one wavefront per query point sweeps ALL faces (lane = face modulo 64:
// exact, no tree to build; ~80 fp64 operations per point-face pair) and reduces
// (squared distance, face index) lexicographically, so ties go to the lowest face index.
// Closest point on a triangle by the Voronoi-region tests of Ericson, "Real-Time
// Collision Detection" 5.1.5; fp64, only + - * /, same operation order as
// the numpy restatement the tests check it against (bit-exact).
// --------------------------------------------------------------------------
__device__ __forceinline__ double dot3d(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

__device__ __forceinline__ void closest_on_triangle(const double* p, const double* a,
                                                    const double* b, const double* c,
                                                    double* q) {
  double ab[3], ac[3], ap[3];
  for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
  const double d1 = dot3d(ab, ap), d2 = dot3d(ac, ap);
  if (d1 <= 0.0 && d2 <= 0.0) { q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; return; }
  double bp[3];
  for (int i = 0; i < 3; ++i) bp[i] = p[i] - b[i];
  const double d3 = dot3d(ab, bp), d4 = dot3d(ac, bp);
  if (d3 >= 0.0 && d4 <= d3) { q[0] = b[0]; q[1] = b[1]; q[2] = b[2]; return; }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
    const double v = d1 / (d1 - d3);
    for (int i = 0; i < 3; ++i) q[i] = a[i] + v * ab[i];
    return;
  }
  double cp[3];
  for (int i = 0; i < 3; ++i) cp[i] = p[i] - c[i];
  const double d5 = dot3d(ab, cp), d6 = dot3d(ac, cp);
  if (d6 >= 0.0 && d5 <= d6) { q[0] = c[0]; q[1] = c[1]; q[2] = c[2]; return; }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
    const double w = d2 / (d2 - d6);
    for (int i = 0; i < 3; ++i) q[i] = a[i] + w * ac[i];
    return;
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
    const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for (int i = 0; i < 3; ++i) q[i] = b[i] + w * (c[i] - b[i]);
    return;
  }
  const double denom = 1.0 / (va + vb + vc);
  const double v = vb * denom, w = vc * denom;
  for (int i = 0; i < 3; ++i) q[i] = a[i] + ab[i] * v + ac[i] * w;
}

__global__ __launch_bounds__(256) void project_to_mesh_kernel(
    const double* __restrict__ pts, int64_t n, const double* __restrict__ verts,
    const int32_t* __restrict__ faces, int64_t nf, double* out, int32_t* face_idx) {
  const int lane = threadIdx.x & 63;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (i >= n) return;                               // wave-uniform
  const double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  double best = INFINITY, bq[3] = {0.0, 0.0, 0.0};
  int64_t bf = nf;
  for (int64_t f = lane; f < nf; f += 64) {
    const int32_t ia = faces[3 * f], ib = faces[3 * f + 1], ic = faces[3 * f + 2];
    const double a[3] = {verts[3 * ia], verts[3 * ia + 1], verts[3 * ia + 2]};
    const double b[3] = {verts[3 * ib], verts[3 * ib + 1], verts[3 * ib + 2]};
    const double c[3] = {verts[3 * ic], verts[3 * ic + 1], verts[3 * ic + 2]};
    double q[3];
    closest_on_triangle(p, a, b, c, q);
    const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    const double d2 = dx * dx + dy * dy + dz * dz;
    if (d2 < best) { best = d2; bf = f; bq[0] = q[0]; bq[1] = q[1]; bq[2] = q[2]; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_xor(best, off, 64);
    const int64_t of = __shfl_xor(bf, off, 64);
    const double o0 = __shfl_xor(bq[0], off, 64), o1 = __shfl_xor(bq[1], off, 64),
                 o2 = __shfl_xor(bq[2], off, 64);
    if (ob < best || (ob == best && of < bf)) {
      best = ob; bf = of; bq[0] = o0; bq[1] = o1; bq[2] = o2;
    }
  }
  if (lane == 0) {
    out[3 * i] = bq[0]; out[3 * i + 1] = bq[1]; out[3 * i + 2] = bq[2];
    if (face_idx) face_idx[i] = static_cast<int32_t>(bf);
  }
}

}  // namespace
}  // namespace epos

using namespace epos;

extern "C" int epos_project_to_mesh_f64(const double* pts, int64_t n,
                                        const double* verts, int64_t nv,
                                        const int32_t* faces, int64_t nf, double* out,
                                        int32_t* face_idx, void* stream) {
  EPOS_REQUIRE(pts && verts && faces && out, "null pointer");
  EPOS_REQUIRE(nv > 0 && nf > 0 && nf < (1LL << 31), "empty mesh");
  if (n == 0) return EPOS_OK;
  hipLaunchKernelGGL(project_to_mesh_kernel,
                     dim3(static_cast<unsigned>(ceil_div(n, 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), pts, n, verts, faces, nf, out,
                     face_idx);
  return launch_status("project_to_mesh_kernel");
}

extern "C" int epos_corr_count(const float* obj_confs, const float* frag_confs,
                               const EposCorrSlot* slots, int S, int B, int P,
                               int O, int F, float min_obj_conf,
                               float min_frag_rel_conf, int32_t* px_off,
                               int32_t* corr_off, uint64_t* frag_mask,
                               int32_t* totals, void* stream) {
  EPOS_REQUIRE(obj_confs && frag_confs && slots && px_off && corr_off &&
               frag_mask && totals, "null pointer");
  EPOS_REQUIRE(F >= 1 && F <= 64, "num_frags must be in [1, 64]");
  EPOS_REQUIRE(B > 0 && P > 0 && O > 0, "empty problem");
  if (S == 0) return EPOS_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(static_cast<unsigned>(ceil_div(P, 4 * CORR_PPW)), S);
  hipLaunchKernelGGL(corr_mask_kernel, grid, dim3(256), 0, st, obj_confs,
                     frag_confs, slots, P, O, F, min_obj_conf, min_frag_rel_conf,
                     px_off, corr_off, frag_mask);
  int rc = launch_status("corr_mask_kernel");
  if (rc) return rc;
  if (P % 4 == 0 && (reinterpret_cast<uintptr_t>(px_off) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(corr_off) & 15) == 0)
    hipLaunchKernelGGL(corr_scan_kernel<4>, dim3(S), dim3(1024), 0, st, px_off,
                       corr_off, P, totals);
  else
    hipLaunchKernelGGL(corr_scan_kernel<1>, dim3(S), dim3(1024), 0, st, px_off,
                       corr_off, P, totals);
  return launch_status("corr_scan_kernel");
}

extern "C" int epos_corr_slot_bases(const int32_t* totals, int S,
                                    int64_t* slot_base, void* stream) {
  EPOS_REQUIRE(totals && slot_base, "null pointer");
  hipLaunchKernelGGL(corr_slot_bases_kernel, dim3(1), dim3(64), 0,
                     static_cast<hipStream_t>(stream), totals, S, slot_base);
  return launch_status("corr_slot_bases_kernel");
}

extern "C" int epos_corr_fill(const float* obj_confs, const float* frag_confs,
                              const float* frag_coords, const double* frag_centers,
                              const double* frag_sizes, const EposCorrSlot* slots,
                              int S, int B, int P, int W, int O, int F,
                              double inv_scale, const int32_t* px_off,
                              const int32_t* corr_off, const uint64_t* frag_mask,
                              const int64_t* slot_base, int64_t capacity,
                              const EposCorrOut* out, int32_t* overflow,
                              void* stream) {
  EPOS_REQUIRE(obj_confs && frag_confs && frag_coords && frag_centers &&
               frag_sizes && slots && px_off && corr_off && frag_mask &&
               slot_base && out && overflow, "null pointer");
  EPOS_REQUIRE(F >= 1 && F <= 64, "num_frags must be in [1, 64]");
  EPOS_REQUIRE(W > 0 && P % W == 0, "P must be a multiple of W");
  if (S == 0) return EPOS_OK;
  dim3 grid(static_cast<unsigned>(ceil_div(P, 4 * CORR_PPW)), S);
  hipLaunchKernelGGL(corr_fill_kernel, grid, dim3(256), 0,
                     static_cast<hipStream_t>(stream), obj_confs, frag_confs,
                     frag_coords, frag_centers, frag_sizes, slots, P, W, O, F,
                     inv_scale, px_off, corr_off, frag_mask, slot_base, capacity,
                     *out, overflow);
  return launch_status("corr_fill_kernel");
}
