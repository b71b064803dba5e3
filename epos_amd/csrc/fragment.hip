// Surface fragmentation by furthest-point sampling on gfx950 -- replaces
// epos_lib/fragment.py:8-54 (fragmentation_fps), the one-off model preprocessing
// behind ObjectModelStore.fragment_models (datagen.py:86-126).
//
// fp64 and the reference's operation order (distance = sqrt((dx*dx + dy*dy) +
// dz*dz), -ffp-contract=off), arg-max/arg-min ties resolved to the lowest index
// like np.argmax / a nearest-centre query without ties, so the result is
// bit-identical to the reference on the golden vectors.
// One workgroup of 1024 threads per call: the FPS loop is inherently sequential
// over the F centres (each needs the previous one), so the parallelism is over
// the vertices; F x 2 barriers in total.
#include "common.h"

namespace epos {
namespace {

__device__ __forceinline__ double dist3(const double* v, double cx, double cy,
                                        double cz) {
  const double dx = v[0] - cx, dy = v[1] - cy, dz = v[2] - cz;
  return sqrt((dx * dx + dy * dy) + dz * dz);
}

__global__ __launch_bounds__(1024) void fps_kernel(const double* __restrict__ verts,
                                                   int64_t V, int F, double* nn,
                                                   double* centers,
                                                   int32_t* center_idx) {
  __shared__ double s_val[1024];
  __shared__ int64_t s_idx[1024];
  __shared__ double s_c[3];
  const int t = threadIdx.x;
  // Distances to the origin: FPS is seeded with the model origin, which is then
  // dropped from the centre list (fragment.py:27-32, 46-47).
  for (int64_t i = t; i < V; i += 1024) nn[i] = dist3(verts + 3 * i, 0.0, 0.0, 0.0);
  __syncthreads();
  for (int f = 0; f < F; ++f) {
    double best = -INFINITY;
    int64_t bi = INT64_MAX;
    for (int64_t i = t; i < V; i += 1024) {
      const double d = nn[i];
      if (d > best) { best = d; bi = i; }           // first maximum wins (np.argmax)
    }
    s_val[t] = best; s_idx[t] = bi;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
      if (t < off) {
        const double v = s_val[t + off];
        const int64_t vi = s_idx[t + off];
        if (v > s_val[t] || (v == s_val[t] && vi < s_idx[t])) { s_val[t] = v; s_idx[t] = vi; }
      }
      __syncthreads();
    }
    const int64_t ci = s_idx[0];
    if (t < 3) {
      const double c = verts[3 * ci + t];
      s_c[t] = c;
      centers[3 * f + t] = c;
    }
    if (t == 0) { center_idx[f] = static_cast<int32_t>(ci); nn[ci] = -1.0; }   // :41
    __syncthreads();
    const double cx = s_c[0], cy = s_c[1], cz = s_c[2];
    for (int64_t i = t; i < V; i += 1024) {
      const double d = dist3(verts + 3 * i, cx, cy, cz);                       // :42-43
      const double o = nn[i];
      nn[i] = d < o ? d : o;                        // np.minimum
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void fps_assign_kernel(
    const double* __restrict__ verts, int64_t V, const double* __restrict__ centers,
    int F, int32_t* frag_ids) {
  extern __shared__ double s_cent[];
  for (int i = threadIdx.x; i < 3 * F; i += blockDim.x) s_cent[i] = centers[i];
  __syncthreads();
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= V) return;
  const double x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
  double best = INFINITY;
  int arg = 0;
  for (int f = 0; f < F; ++f) {
    const double dx = x - s_cent[3 * f], dy = y - s_cent[3 * f + 1],
                 dz = z - s_cent[3 * f + 2];
    const double d = (dx * dx + dy * dy) + dz * dz;
    if (d < best) { best = d; arg = f; }
  }
  frag_ids[i] = arg;
}

}  // namespace
}  // namespace epos

extern "C" int epos_fragmentation_fps(const double* vertices, int64_t V,
                                      int num_frags, double* nn_dist,
                                      double* centers, int32_t* center_idx,
                                      int32_t* vertex_frag_ids, void* stream) {
  using namespace epos;
  EPOS_REQUIRE(vertices && nn_dist && centers && center_idx && vertex_frag_ids,
               "null pointer");
  EPOS_REQUIRE(V >= num_frags && num_frags >= 1, "need at least num_frags vertices");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(fps_kernel, dim3(1), dim3(1024), 0, st, vertices, V, num_frags,
                     nn_dist, centers, center_idx);
  int rc = launch_status("fps_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(fps_assign_kernel, dim3(static_cast<unsigned>(ceil_div(V, 256))),
                     dim3(256), sizeof(double) * 3 * num_frags, st, vertices, V,
                     centers, num_frags, vertex_frag_ids);
  return launch_status("fps_assign_kernel");
}
