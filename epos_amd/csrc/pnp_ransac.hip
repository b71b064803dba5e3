// Wavefront-parallel multi-instance PnP-RANSAC for gfx950 -- replaces the
// pyprogressivex.find6DPoses call of scripts/infer.py:470-488 (the un-vendored
// danini/progressive-x C++ module). Algorithm definition: DESIGN.md "Pose
// fitting" (P3P minimal solver, MSAC quality, Gauss-Newton local optimisation of
// the best hypothesis, sequential multi-instance with Tanimoto / coverage tests).
//
// Mapping to the hardware:
//   * ransac_hypotheses: ONE HYPOTHESIS SET PER WAVEFRONT. The 64 lanes compute
//     the (wave-uniform) P3P solve redundantly, then stride over the slot's
//     correspondences; inliers are counted with popcount, the MSAC sum is reduced
//     with a fixed xor-butterfly so the result does not depend on scheduling.
//     Grid = (max_iters / 4, slots): all objects of all images in one launch.
//   * ransac_select_refine: one workgroup per slot: deterministic arg-max over the
//     hypothesis table, then wave 0 runs the local optimisation (27 normal-
//     equation sums per Gauss-Newton step, butterfly-reduced), the instance
//     acceptance tests on inlier bitsets (ballot + popcount) and the stable
//     compaction of the still-unexplained correspondences.
// Arithmetic is fp64 (the reference hands f64 arrays to progressive-x) using only
// + - * / sqrt, compiled with -ffp-contract=off: results are reproducible run to
// run and identical to a scalar evaluation in the same canonical order.
#include "common.h"

namespace epos {
namespace {

constexpr int MAX_SOL = 4;

// ------------------------------------------------------------------ RNG --
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t draw(uint64_t seed, uint32_t round,
                                         uint32_t it, uint32_t k, uint64_t n) {
  const uint64_t u =
      mix64(mix64(seed ^ mix64((static_cast<uint64_t>(round) << 32) | it)) + k);
  return __umul64hi(u, n);
}
__device__ __forceinline__ void sample3(uint64_t seed, uint32_t round, uint32_t it,
                                        int64_t m, int64_t* s) {
  int64_t a = static_cast<int64_t>(draw(seed, round, it, 0, m));
  int64_t b = static_cast<int64_t>(draw(seed, round, it, 1, m - 1));
  int64_t c = static_cast<int64_t>(draw(seed, round, it, 2, m - 2));
  if (b >= a) b += 1;
  const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
  if (c >= lo) c += 1;
  if (c >= hi) c += 1;
  s[0] = a; s[1] = b; s[2] = c;
}

// --------------------------------------------------------- small algebra --
__device__ __forceinline__ double dot3(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
// symmetric 3x3 packed as {00, 01, 02, 11, 12, 22}
__device__ __forceinline__ void sym_adj(const double* s, double* b) {
  b[0] = s[3] * s[5] - s[4] * s[4];
  b[1] = s[2] * s[4] - s[1] * s[5];
  b[2] = s[1] * s[4] - s[2] * s[3];
  b[3] = s[0] * s[5] - s[2] * s[2];
  b[4] = s[1] * s[2] - s[0] * s[4];
  b[5] = s[0] * s[3] - s[1] * s[1];
}
__device__ __forceinline__ double sym_det(const double* s, const double* adj) {
  return s[0] * adj[0] + s[1] * adj[1] + s[2] * adj[2];
}
__device__ __forceinline__ double sym_inner(const double* a, const double* b) {
  return a[0] * b[0] + a[3] * b[3] + a[5] * b[5] +
         2.0 * (a[1] * b[1] + a[2] * b[2] + a[4] * b[4]);
}
__device__ __forceinline__ double sym_quad(const double* s, const double* v) {
  return s[0] * v[0] * v[0] + s[3] * v[1] * v[1] + s[5] * v[2] * v[2] +
         2.0 * (s[1] * v[0] * v[1] + s[2] * v[0] * v[2] + s[4] * v[1] * v[2]);
}
__device__ __forceinline__ double sym_at(const double* s, int i, int j) {
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  // (0,0)->0 (0,1)->1 (0,2)->2 (1,1)->3 (1,2)->4 (2,2)->5
  return s[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
}

__device__ __forceinline__ double cubic_eval(double x, double b, double c, double d) {
  return ((x + b) * x + c) * x + d;
}

// One real root of x^3 + b x^2 + c x + d: Newton started beyond the outer
// turning point (monotone convergence).
__device__ double cubic_root(double b, double c, double d) {
  double x;
  const double disc = b * b - 3.0 * c;
  if (disc > 0.0) {
    const double v = sqrt(disc);
    const double t1 = (-b - v) / 3.0;
    const double f1 = cubic_eval(t1, b, c, d);
    if (f1 > 0.0) {
      double step = v / 3.0 + 1e-3;
      x = t1 - step;
      for (int i = 0; i < 200 && cubic_eval(x, b, c, d) > 0.0; ++i) { step *= 2.0; x = t1 - step; }
    } else {
      const double t2 = (-b + v) / 3.0;
      double step = v / 3.0 + 1e-3;
      x = t2 + step;
      for (int i = 0; i < 200 && cubic_eval(x, b, c, d) < 0.0; ++i) { step *= 2.0; x = t2 + step; }
    }
  } else {
    const double t0 = -b / 3.0;
    const double f0 = cubic_eval(t0, b, c, d);
    double step = 1.0 + fabs(t0);
    x = t0;
    if (f0 > 0.0) {
      x = t0 - step;
      for (int i = 0; i < 200 && cubic_eval(x, b, c, d) > 0.0; ++i) { step *= 2.0; x = t0 - step; }
    } else if (f0 < 0.0) {
      x = t0 + step;
      for (int i = 0; i < 200 && cubic_eval(x, b, c, d) < 0.0; ++i) { step *= 2.0; x = t0 + step; }
    }
  }
  for (int i = 0; i < 60; ++i) {
    const double f = cubic_eval(x, b, c, d);
    const double fp = (3.0 * x + 2.0 * b) * x + c;
    if (fp == 0.0) break;
    const double dx = f / fp;
    x -= dx;
    if (fabs(dx) <= 1e-15 * fabs(x)) break;
  }
  return x;
}

// ------------------------------------------------------------------ P3P --
// f[9]: three unit bearings (row major), X[9]: three world points. Writes up to
// 4 poses (R row-major 9 + t 3) to pose[k*12]; returns their number.
__device__ int p3p(const double* f, const double* X, double* pose) {
  const double c12 = dot3(f, f + 3), c13 = dot3(f, f + 6), c23 = dot3(f + 3, f + 6);
  double d12[3], d13[3], d23[3];
  for (int i = 0; i < 3; ++i) {
    d12[i] = X[i] - X[3 + i];
    d13[i] = X[i] - X[6 + i];
    d23[i] = X[3 + i] - X[6 + i];
  }
  const double a12 = dot3(d12, d12), a13 = dot3(d13, d13), a23 = dot3(d23, d23);
  double nx[3];
  cross3(d12, d13, nx);
  const double detx = dot3(nx, nx);
  if (!(detx > 1e-18 * (a12 * a13 + 1e-300))) return 0;
  double r0[3], r1[3];
  cross3(d13, nx, r0);
  cross3(nx, d12, r1);
  double Xinv[9];
  for (int i = 0; i < 3; ++i) {
    Xinv[i] = r0[i] / detx;
    Xinv[3 + i] = r1[i] / detx;
    Xinv[6 + i] = nx[i] / detx;
  }
  const double M12[6] = {1, -c12, 0, 1, 0, 0};
  const double M13[6] = {1, 0, -c13, 0, 0, 1};
  const double M23[6] = {0, 0, 0, 1, -c23, 1};
  double D1[6], D2[6];
  for (int i = 0; i < 6; ++i) {
    D1[i] = a23 * M12[i] - a12 * M23[i];
    D2[i] = a23 * M13[i] - a13 * M23[i];
  }
  double A1[6], A2[6];
  sym_adj(D1, A1);
  sym_adj(D2, A2);
  const double k0 = sym_det(D1, A1), k3 = sym_det(D2, A2);
  const double k1 = sym_inner(A1, D2), k2 = sym_inner(A2, D1);
  double D0[6], E[6];
  if (fabs(k3) >= fabs(k0)) {
    if (k3 == 0.0) return 0;
    const double g = cubic_root(k2 / k3, k1 / k3, k0 / k3);
    const bool use2 = fabs(g) <= 1.0;
    for (int i = 0; i < 6; ++i) { D0[i] = D1[i] + g * D2[i]; E[i] = use2 ? D2[i] : D1[i]; }
  } else {
    const double g = cubic_root(k1 / k0, k2 / k0, k3 / k0);
    const bool use1 = fabs(g) <= 1.0;
    for (int i = 0; i < 6; ++i) { D0[i] = g * D1[i] + D2[i]; E[i] = use1 ? D1[i] : D2[i]; }
  }
  double B[6];
  sym_adj(D0, B);
  int bi = 0;
  double bmax = -B[0];
  if (-B[3] > bmax) { bmax = -B[3]; bi = 1; }
  if (-B[5] > bmax) { bmax = -B[5]; bi = 2; }
  if (!(bmax > 0.0)) return 0;
  const double sq = sqrt(bmax);
  double pt[3];
  for (int i = 0; i < 3; ++i) pt[i] = -sym_at(B, i, bi) / sq;
  double N[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) N[i * 3 + j] = sym_at(D0, i, j);
  N[1] -= pt[2]; N[2] += pt[1];
  N[3] += pt[2]; N[5] -= pt[0];
  N[6] -= pt[1]; N[7] += pt[0];
  int ri = 0, ci = 0;
  double rbest = -1.0, cbest = -1.0;
  for (int i = 0; i < 3; ++i) {
    const double rn = N[i * 3] * N[i * 3] + N[i * 3 + 1] * N[i * 3 + 1] + N[i * 3 + 2] * N[i * 3 + 2];
    const double cn = N[i] * N[i] + N[3 + i] * N[3 + i] + N[6 + i] * N[6 + i];
    if (rn > rbest) { rbest = rn; ri = i; }
    if (cn > cbest) { cbest = cn; ci = i; }
  }
  double planes[6];
  for (int j = 0; j < 3; ++j) { planes[j] = N[ri * 3 + j]; planes[3 + j] = N[j * 3 + ci]; }

  int nsol = 0;
  for (int pl = 0; pl < 2; ++pl) {
    const double* n = planes + 3 * pl;
    int k = 0;
    if (fabs(n[1]) > fabs(n[k])) k = 1;
    if (fabs(n[2]) > fabs(n[k])) k = 2;
    if (n[k] == 0.0) continue;
    const int a = (k + 1) % 3, b = (k + 2) % 3;
    const double ua = -n[a] / n[k], ub = -n[b] / n[k];
    const double Eaa = sym_at(E, a, a), Ebb = sym_at(E, b, b), Ekk = sym_at(E, k, k);
    const double Eab = sym_at(E, a, b), Eak = sym_at(E, a, k), Ebk = sym_at(E, b, k);
    const double q00 = Eaa + 2.0 * ua * Eak + ua * ua * Ekk;
    const double q11 = Ebb + 2.0 * ub * Ebk + ub * ub * Ekk;
    const double q01 = Eab + ub * Eak + ua * Ebk + ua * ub * Ekk;
    const double disc = q01 * q01 - q00 * q11;
    if (!(disc >= 0.0)) continue;
    const double sd = sqrt(disc);
    for (int sg = 0; sg < 2; ++sg) {
      double la, lb;
      const double num = sg == 0 ? (-q01 + sd) : (-q01 - sd);
      if (fabs(q00) >= fabs(q11)) {
        if (q00 == 0.0) continue;
        la = num / q00; lb = 1.0;
      } else {
        la = 1.0; lb = num / q11;
      }
      double lam[3];
      const double lk = ua * la + ub * lb;
      lam[0] = a == 0 ? la : (b == 0 ? lb : lk);
      lam[1] = a == 1 ? la : (b == 1 ? lb : lk);
      lam[2] = a == 2 ? la : (b == 2 ? lb : lk);
      const double qs = sym_quad(M12, lam) + sym_quad(M13, lam) + sym_quad(M23, lam);
      if (!(qs > 0.0)) continue;
      double sc = sqrt((a12 + a13 + a23) / qs);
      if (lam[0] < 0.0) sc = -sc;
      lam[0] *= sc; lam[1] *= sc; lam[2] *= sc;
      if (!(lam[0] > 0.0 && lam[1] > 0.0 && lam[2] > 0.0)) continue;
      double Y[9];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Y[i * 3 + j] = lam[i] * f[i * 3 + j];
      double e12[3], e13[3], ny[3];
      for (int j = 0; j < 3; ++j) { e12[j] = Y[j] - Y[3 + j]; e13[j] = Y[j] - Y[6 + j]; }
      cross3(e12, e13, ny);
      double* R = pose + nsol * 12;
      double* t = R + 9;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          R[i * 3 + j] = e12[i] * Xinv[j] + e13[i] * Xinv[3 + j] + ny[i] * Xinv[6 + j];
      for (int i = 0; i < 3; ++i)
        t[i] = Y[i] - (R[i * 3] * X[0] + R[i * 3 + 1] * X[1] + R[i * 3 + 2] * X[2]);
      ++nsol;
    }
  }
  return nsol;
}

// -------------------------------------------------------------- scoring --
__device__ __forceinline__ double butterfly_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ int butterfly_sum_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// returns true if behind the camera; else e2 / Xc / r are set
__device__ __forceinline__ bool reproj(const double* pose, const double* K,
                                       const double* xy, const double* xyz,
                                       double* e2, double* Xc, double* r) {
  Xc[0] = pose[0] * xyz[0] + pose[1] * xyz[1] + pose[2] * xyz[2] + pose[9];
  Xc[1] = pose[3] * xyz[0] + pose[4] * xyz[1] + pose[5] * xyz[2] + pose[10];
  Xc[2] = pose[6] * xyz[0] + pose[7] * xyz[1] + pose[8] * xyz[2] + pose[11];
  if (!(Xc[2] > 0.0)) return true;
  const double iz = 1.0 / Xc[2];               // the only division per point
  const double px = (K[0] * Xc[0] + K[1] * Xc[1]) * iz + K[2];
  const double py = (K[4] * Xc[1]) * iz + K[5];
  r[0] = px - xy[0];
  r[1] = py - xy[1];
  *e2 = r[0] * r[0] + r[1] * r[1];
  return false;
}

constexpr int PF = 4;           // items fetched together per lane / thread

// PF correspondences of one lane's strided partial (items i0, i0 + stride, ...): all
// index loads first, then all point loads (unconditional, from clamped positions), so
// that the gathers of PF items are in flight together.
struct PointBatch {
  double x2[PF][2], x3[PF][3];
  bool ok[PF];
  __device__ __forceinline__ void load(const double* xy, const double* xyz,
                                       const int32_t* idx, int64_t i0, int stride,
                                       int64_t m) {
    int32_t p[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int64_t i = i0 + static_cast<int64_t>(u) * stride;
      ok[u] = i < m;
      p[u] = idx[ok[u] ? i : i0];
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      x2[u][0] = xy[2 * p[u]]; x2[u][1] = xy[2 * p[u] + 1];
      x3[u][0] = xyz[3 * p[u]]; x3[u][1] = xyz[3 * p[u] + 1]; x3[u][2] = xyz[3 * p[u] + 2];
    }
  }
};

// MSAC scores + inlier counts of `ns` (<= 4) poses over idx[0..m) in ONE pass over
// the correspondences (wave-wide; results uniform). Each pose's sum keeps the
// canonical order: 64 strided per-lane partials, then the xor butterfly.
__device__ void score_poses(const double* poses, int ns, const double* K,
                            const double* xy, const double* xyz, const int32_t* idx,
                            int64_t m, double thr2, int lane, double* score,
                            int* count) {
  const double inv_thr2 = 1.0 / thr2;
  double acc[MAX_SOL] = {0.0, 0.0, 0.0, 0.0};
  int cnt[MAX_SOL] = {0, 0, 0, 0};
  // PF items of this lane's partial are fetched together (index -> point is a dependent
  // gather: without this the loop is one memory round trip per item) and then consumed
  // in the canonical order i, i + 64, ...
  for (int64_t i0 = lane; i0 < m; i0 += 64 * PF) {
    PointBatch pb;
    pb.load(xy, xyz, idx, i0, 64, m);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (!pb.ok[u]) continue;
#pragma unroll
      for (int q = 0; q < MAX_SOL; ++q) {
        if (q < ns) {                                      // wave-uniform
          double e2, Xc[3], r[2];
          if (!reproj(poses + 12 * q, K, pb.x2[u], pb.x3[u], &e2, Xc, r) && e2 < thr2) {
            acc[q] += 1.0 - e2 * inv_thr2;
            ++cnt[q];
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < MAX_SOL; ++q) {
    if (q < ns) {
      count[q] = butterfly_sum_i(cnt[q]);
      score[q] = butterfly_sum(acc[q]);
    }
  }
}

__device__ void bearing(const double* K, const double* xy, double* f) {
  const double y = (xy[1] - K[5]) / K[4];
  const double x = (xy[0] - K[2] - K[1] * y) / K[0];
  const double n = sqrt(x * x + y * y + 1.0);
  f[0] = x / n; f[1] = y / n; f[2] = 1.0 / n;
}

struct Work {
  double* hyp_score;     // [S][iters][4]
  double* hyp_pose;      // [S][iters][4][12]
  int32_t* hyp_count;    // [S][iters][4]
  int32_t* active;       // [N] local indices, pooled by slot_base
  int32_t* n_active;     // [S]
  int32_t* done;         // [S]
  uint64_t* inl_bits;    // [max_k][words_total]
  int64_t words_total;
};

__global__ __launch_bounds__(256) void ransac_init(const int64_t* slot_base, int S,
                                                   Work w, int32_t* labels,
                                                   int32_t* num_models,
                                                   int min_pts, int64_t n_capacity) {
  const int s = blockIdx.x;
  const int64_t base = slot_base[s];
  // A slot whose rows would end beyond the pooled arrays (the correspondence stage
  // raised its overflow flag and wrote nothing there) is fitted as EMPTY: no kernel of
  // this stage then touches a row >= n_capacity. The host reports the overflow.
  const int64_t n = slot_base[s + 1] <= n_capacity ? slot_base[s + 1] - base : 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    w.active[base + i] = static_cast<int32_t>(i);
    labels[base + i] = -1;
  }
  if (threadIdx.x == 0) {
    w.n_active[s] = static_cast<int32_t>(n);
    num_models[s] = 0;
    w.done[s] = (n < min_pts || n < 3) ? 1 : 0;      // infer.py:420-422
  }
}

__global__ __launch_bounds__(256) void ransac_hypotheses(
    const double* __restrict__ xy_all, const double* __restrict__ xyz_all,
    const int64_t* __restrict__ slot_base, const double* __restrict__ Ks,
    const uint64_t* __restrict__ seeds, const int32_t* __restrict__ max_models,
    const int32_t* __restrict__ num_models, EposFitParams prm, int max_k, int round,
    Work w) {
  const int s = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (it >= prm.max_iters) return;
  double* hs = w.hyp_score + (static_cast<int64_t>(s) * prm.max_iters + it) * MAX_SOL;
  int ns = 0;
  if (!w.done[s]) {
    const int64_t base = slot_base[s];
    const double* xy = xy_all + 2 * base;
    const double* xyz = xyz_all + 3 * base;
    const int32_t* active = w.active + base;
    const int64_t n_active = w.n_active[s];
    double K[9];
    for (int i = 0; i < 9; ++i) K[i] = Ks[s * 9 + i];
    int64_t m = n_active;
    if (prm.use_prosac) {
      m = (n_active * static_cast<int64_t>(it + 1) + prm.max_iters - 1) / prm.max_iters;
      if (m < prm.min_point_number) m = prm.min_point_number;
      if (m < 3) m = 3;
      if (m > n_active) m = n_active;
    }
    int64_t smp[3];
    sample3(seeds[s], static_cast<uint32_t>(round), static_cast<uint32_t>(it), m, smp);
    double f[9], X[9], p2[6];
    for (int j = 0; j < 3; ++j) {
      const int32_t p = active[smp[j]];
      bearing(K, xy + 2 * p, f + 3 * j);
      for (int d = 0; d < 3; ++d) X[3 * j + d] = xyz[3 * p + d];
      p2[2 * j] = xy[2 * p]; p2[2 * j + 1] = xy[2 * p + 1];
    }
    const double area = 0.5 * fabs((p2[2] - p2[0]) * (p2[5] - p2[1]) -
                                   (p2[3] - p2[1]) * (p2[4] - p2[0]));
    if (!(area < prm.min_triangle_area)) {
      double sols[MAX_SOL * 12];
      ns = p3p(f, X, sols);
      const double thr2 = prm.threshold * prm.threshold;
      double* hp = w.hyp_pose + (static_cast<int64_t>(s) * prm.max_iters + it) * MAX_SOL * 12;
      int32_t* hc = w.hyp_count + (static_cast<int64_t>(s) * prm.max_iters + it) * MAX_SOL;
      double sc[MAX_SOL];
      int cnt[MAX_SOL];
      score_poses(sols, ns, K, xy, xyz, active, n_active, thr2, lane, sc, cnt);
#pragma unroll
      for (int q = 0; q < MAX_SOL; ++q) {
        if (q < ns) {
          if (lane == 0) { hs[q] = sc[q]; hc[q] = cnt[q]; }
#pragma unroll
          for (int e = 0; e < 12; ++e)
            if (lane == e) hp[q * 12 + e] = sols[12 * q + e];
        }
      }
    }
  }
  if (lane == 0)
    for (int q = ns; q < MAX_SOL; ++q) hs[q] = -1.0;     // no hypothesis
}

// ------------------------------------------------- local optimisation --
__device__ void orthonormalize(double* R) {
  const double n0 = sqrt(dot3(R, R));
  for (int j = 0; j < 3; ++j) R[j] /= n0;
  const double d = dot3(R, R + 3);
  for (int j = 0; j < 3; ++j) R[3 + j] -= d * R[j];
  const double n1 = sqrt(dot3(R + 3, R + 3));
  for (int j = 0; j < 3; ++j) R[3 + j] /= n1;
  cross3(R, R + 3, R + 6);
}

__device__ int solve6(const double* H /*[36]*/, const double* g, double* x) {
  double A[6][7];
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) A[i][j] = H[i * 6 + j];
    A[i][6] = -g[i];
  }
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    for (int r = c + 1; r < 6; ++r)
      if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
    if (!(fabs(A[piv][c]) > 1e-300)) return 1;
    if (piv != c)
      for (int j = 0; j < 7; ++j) { const double tmp = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = tmp; }
    for (int r = c + 1; r < 6; ++r) {
      const double fct = A[r][c] / A[c][c];
      for (int j = c; j < 7; ++j) A[r][j] -= fct * A[c][j];
    }
  }
  for (int i = 5; i >= 0; --i) {
    double sacc = A[i][6];
    for (int j = i + 1; j < 6; ++j) sacc -= A[i][j] * x[j];
    x[i] = sacc / A[i][i];
  }
  return 0;
}

// ---- workgroup-wide (256 threads) sums of the local optimisation -------------------
// Canonical order: 256 strided partials (thread t takes items t, t+256, ...), the xor
// butterfly inside each wave, then (w0 + w1) + (w2 + w3) through LDS. Every thread gets
// the same value, so the control flow that depends on it stays workgroup-uniform.
__device__ __forceinline__ double block_combine(double wave_sum, double* s_red, int t) {
  if ((t & 63) == 0) s_red[t >> 6] = wave_sum;
  __syncthreads();
  const double r = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  __syncthreads();
  return r;
}

__device__ double score_pose_block(const double* pose, const double* K, const double* xy,
                                   const double* xyz, const int32_t* idx, int64_t m,
                                   double thr2, int t, double* s_red, int* s_cnt,
                                   int* count) {
  const double inv_thr2 = 1.0 / thr2;
  double acc = 0.0;
  int cnt = 0;
  for (int64_t i0 = t; i0 < m; i0 += 256 * PF) {
    PointBatch pb;
    pb.load(xy, xyz, idx, i0, 256, m);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (!pb.ok[u]) continue;
      double e2, Xc[3], r[2];
      if (reproj(pose, K, pb.x2[u], pb.x3[u], &e2, Xc, r)) continue;
      if (e2 < thr2) { acc += 1.0 - e2 * inv_thr2; ++cnt; }
    }
  }
  cnt = butterfly_sum_i(cnt);
  if ((t & 63) == 0) s_cnt[t >> 6] = cnt;
  const double sc = block_combine(butterfly_sum(acc), s_red, t);   // two barriers inside
  *count = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  __syncthreads();
  return sc;
}

__device__ int gn_step_block(const double* pose, const double* K, const double* xy,
                             const double* xyz, const int32_t* idx, int64_t m,
                             double thr2, int t, double* s_red27 /*[4][27]*/,
                             double* next) {
  double acc[27];
#pragma unroll
  for (int v = 0; v < 27; ++v) acc[v] = 0.0;
  for (int64_t i0 = t; i0 < m; i0 += 256 * PF) {
    PointBatch pb;
    pb.load(xy, xyz, idx, i0, 256, m);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
    if (!pb.ok[u]) continue;
    double e2, Xc[3], r[2];
    if (reproj(pose, K, pb.x2[u], pb.x3[u], &e2, Xc, r)) continue;
    if (!(e2 < thr2)) continue;
    const double iz = 1.0 / Xc[2];
    const double a0 = K[0] * iz, a1 = K[1] * iz,
                 a2 = -(K[0] * Xc[0] + K[1] * Xc[1]) * iz * iz;
    const double b1 = K[4] * iz, b2 = -(K[4] * Xc[1]) * iz * iz;
    double J0[6], J1[6];
    J0[0] = a1 * Xc[2] - a2 * Xc[1];
    J0[1] = -a0 * Xc[2] + a2 * Xc[0];
    J0[2] = a0 * Xc[1] - a1 * Xc[0];
    J0[3] = a0; J0[4] = a1; J0[5] = a2;
    J1[0] = b1 * Xc[2] - b2 * Xc[1];
    J1[1] = b2 * Xc[0];
    J1[2] = -b1 * Xc[0];
    J1[3] = 0.0; J1[4] = b1; J1[5] = b2;
    int v = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 6; ++b) { acc[v] += J0[a] * J0[b] + J1[a] * J1[b]; ++v; }
#pragma unroll
    for (int a = 0; a < 6; ++a) { acc[v] += J0[a] * r[0] + J1[a] * r[1]; ++v; }
    }
  }
#pragma unroll
  for (int v = 0; v < 27; ++v) {
    const double ws = butterfly_sum(acc[v]);
    if ((t & 63) == 0) s_red27[(t >> 6) * 27 + v] = ws;
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < 27; ++v)
    acc[v] = (s_red27[v] + s_red27[27 + v]) + (s_red27[54 + v] + s_red27[81 + v]);
  __syncthreads();
  double H[36], g[6], x[6];
  int v = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) { H[a * 6 + b] = acc[v]; H[b * 6 + a] = acc[v]; ++v; }
  for (int a = 0; a < 6; ++a) g[a] = acc[v++];
  if (solve6(H, g, x)) return 1;
  double qw = 1.0, qx = 0.5 * x[0], qy = 0.5 * x[1], qz = 0.5 * x[2];
  const double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  qw /= qn; qx /= qn; qy /= qn; qz /= qn;
  double dR[9];
  dR[0] = 1.0 - 2.0 * (qy * qy + qz * qz); dR[1] = 2.0 * (qx * qy - qz * qw); dR[2] = 2.0 * (qx * qz + qy * qw);
  dR[3] = 2.0 * (qx * qy + qz * qw); dR[4] = 1.0 - 2.0 * (qx * qx + qz * qz); dR[5] = 2.0 * (qy * qz - qx * qw);
  dR[6] = 2.0 * (qx * qz - qy * qw); dR[7] = 2.0 * (qy * qz + qx * qw); dR[8] = 1.0 - 2.0 * (qx * qx + qy * qy);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      next[i * 3 + j] = dR[i * 3] * pose[j] + dR[i * 3 + 1] * pose[3 + j] + dR[i * 3 + 2] * pose[6 + j];
    next[9 + i] = dR[i * 3] * pose[9] + dR[i * 3 + 1] * pose[10] + dR[i * 3 + 2] * pose[11] + x[3 + i];
  }
  for (int i = 0; i < 12; ++i)
    if (!(next[i] == next[i])) return 1;
  return 0;
}

__global__ __launch_bounds__(256) void ransac_select_refine(
    const double* __restrict__ xy_all, const double* __restrict__ xyz_all,
    const int64_t* __restrict__ slot_base, const double* __restrict__ Ks,
    const int32_t* __restrict__ max_models, EposFitParams prm, int max_k, Work w,
    double* poses, double* scores, int32_t* num_models, int32_t* labels_all) {
  __shared__ double s_score[256];
  __shared__ int s_index[256];
  const int s = blockIdx.x;
  const int t = threadIdx.x;
  if (w.done[s]) return;                                   // block-uniform
  int want = max_models[s];
  if (want < 0 || want > max_k) want = max_k;
  const int k = num_models[s];
  const int64_t base = slot_base[s];
  const int64_t n = slot_base[s + 1] - base;
  int32_t* active = w.active + base;
  const int64_t n_active = w.n_active[s];
  if (k >= want || n_active < prm.min_point_number || n_active < 3) {
    if (t == 0) w.done[s] = 1;
    return;
  }
  // ---- arg-max over the hypothesis table (ties -> lowest index) ----
  const int nh = prm.max_iters * MAX_SOL;
  const double* hs = w.hyp_score + static_cast<int64_t>(s) * nh;
  double best = -1.0;
  int best_i = 0x7fffffff;
  for (int i = t; i < nh; i += 256) {
    const double v = hs[i];
    if (v > best) { best = v; best_i = i; }
  }
  s_score[t] = best; s_index[t] = best_i;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) {
      const double v = s_score[t + off];
      const int vi = s_index[t + off];
      if (v > s_score[t] || (v == s_score[t] && vi < s_index[t])) {
        s_score[t] = v; s_index[t] = vi;
      }
    }
    __syncthreads();
  }
  __shared__ double s_red[4];
  __shared__ double s_red27[4 * 27];
  __shared__ int s_cnt[4];
  __shared__ int s_inl[4], s_new[4];
  const int lane = t & 63, wave = t >> 6;
  double best_score = s_score[0];
  const int bi = s_index[0];
  if (!(best_score > 0.0)) { if (t == 0) w.done[s] = 1; return; }       // uniform
  int best_count = w.hyp_count[static_cast<int64_t>(s) * nh + bi];
  if (best_count < 3) { if (t == 0) w.done[s] = 1; return; }
  const double* xy = xy_all + 2 * base;
  const double* xyz = xyz_all + 3 * base;
  int32_t* labels = labels_all + base;
  double K[9];
  for (int i = 0; i < 9; ++i) K[i] = Ks[s * 9 + i];
  const double thr2 = prm.threshold * prm.threshold;
  double pose[12];
  for (int i = 0; i < 12; ++i) pose[i] = w.hyp_pose[(static_cast<int64_t>(s) * nh + bi) * 12 + i];
  // ---- local optimisation: the whole workgroup sums, every thread steps ----
  orthonormalize(pose);
  best_score = score_pose_block(pose, K, xy, xyz, active, n_active, thr2, t, s_red, s_cnt,
                                &best_count);
  for (int li = 0; li < prm.lo_iters; ++li) {
    double cand[12];
    if (gn_step_block(pose, K, xy, xyz, active, n_active, thr2, t, s_red27, cand)) break;
    int cnt;
    const double sc = score_pose_block(cand, K, xy, xyz, active, n_active, thr2, t, s_red,
                                       s_cnt, &cnt);
    if (!(sc > best_score)) break;
    best_score = sc; best_count = cnt;
    for (int i = 0; i < 12; ++i) pose[i] = cand[i];
  }
  if (best_count < prm.min_point_number) { if (t == 0) w.done[s] = 1; return; }
  // ---- inliers over ALL correspondences of the slot -> bitset (chunk c: wave c % 4) --
  const int64_t words = (n + 63) / 64;
  const int64_t wbase = base / 64 + s;
  uint64_t* cur = w.inl_bits + static_cast<int64_t>(k) * w.words_total + wbase;
  int n_inl = 0, n_new = 0;
  for (int64_t c = wave; c < words; c += 4) {
    const int64_t i = c * 64 + lane;
    bool inl = false, fresh = false;
    if (i < n) {
      double e2, Xc[3], r[2];
      if (!reproj(pose, K, xy + 2 * i, xyz + 3 * i, &e2, Xc, r)) inl = e2 < thr2;
      fresh = inl && labels[i] < 0;
    }
    const uint64_t bits = __ballot(inl);
    n_inl += __popcll(bits);
    n_new += __popcll(__ballot(fresh));
    if (lane == 0) cur[c] = bits;
  }
  if (lane == 0) { s_inl[wave] = n_inl; s_new[wave] = n_new; }
  __threadfence_block();
  __syncthreads();
  if (t >= 64) return;                                     // wave 0 finishes the round
  n_inl = s_inl[0] + s_inl[1] + s_inl[2] + s_inl[3];
  n_new = s_new[0] + s_new[1] + s_new[2] + s_new[3];
  __threadfence_block();
  bool ok = n_inl > 0;
  for (int j = 0; j < k && ok; ++j) {
    const uint64_t* pj = w.inl_bits + static_cast<int64_t>(j) * w.words_total + wbase;
    int inter = 0, uni = 0;
    for (int64_t c = lane; c < words; c += 64) {
      const uint64_t a = cur[c], b = pj[c];
      inter += __popcll(a & b);
      uni += __popcll(a | b);
    }
    inter = butterfly_sum_i(inter);
    uni = butterfly_sum_i(uni);
    if (static_cast<double>(inter) >= prm.max_tanimoto_similarity * static_cast<double>(uni)) ok = false;
  }
  if (ok && static_cast<double>(n_new) < prm.min_coverage * static_cast<double>(n_inl)) ok = false;
  if (!ok) { if (lane == 0) w.done[s] = 1; return; }
  // ---- accept: write the pose, label + remove its inliers (stable compaction) --
  if (lane < 12) poses[(static_cast<int64_t>(s) * max_k + k) * 12 + lane] = pose[lane];
  if (lane == 0) scores[static_cast<int64_t>(s) * max_k + k] = best_score;
  int64_t wpos = 0;
  for (int64_t c0 = 0; c0 < n_active; c0 += 64) {
    const int64_t i = c0 + lane;
    int32_t p = 0;
    bool keep = false;
    if (i < n_active) {
      p = active[i];
      const bool inl = (cur[p >> 6] >> (p & 63)) & 1ull;
      if (inl) labels[p] = k;
      keep = !inl;
    }
    const uint64_t kb = __ballot(keep);
    const int rank = __popcll(kb & ((1ull << lane) - 1ull));
    if (keep) active[wpos + rank] = p;
    wpos += __popcll(kb);
  }
  if (lane == 0) {
    w.n_active[s] = static_cast<int32_t>(wpos);
    num_models[s] = k + 1;
    if (k + 1 >= want) w.done[s] = 1;
  }
}

inline int64_t align_up(int64_t x) { return (x + 255) / 256 * 256; }

struct Layout {
  int64_t hyp_score, hyp_pose, hyp_count, active, n_active, done, inl_bits, total;
  int64_t words_total;
};

Layout make_layout(int S, int64_t n_cap, int max_iters, int max_k) {
  Layout L;
  int64_t off = 0;
  const int64_t nh = static_cast<int64_t>(S) * max_iters * MAX_SOL;
  L.hyp_score = off; off = align_up(off + nh * 8);
  L.hyp_pose = off; off = align_up(off + nh * 12 * 8);
  L.hyp_count = off; off = align_up(off + nh * 4);
  L.active = off; off = align_up(off + (n_cap + 1) * 4);
  L.n_active = off; off = align_up(off + (S + 1) * 4);
  L.done = off; off = align_up(off + (S + 1) * 4);
  L.words_total = n_cap / 64 + S + 2;
  L.inl_bits = off; off = align_up(off + L.words_total * (max_k + 1) * 8);
  L.total = off;
  return L;
}

}  // namespace
}  // namespace epos

using namespace epos;

extern "C" void epos_fit_params_default(EposFitParams* p) {
  // scripts/infer.py:76-120 (flag defaults) and :470-488 (call).
  p->threshold = 4.0;
  p->neighborhood_ball_radius = 20.0;
  p->spatial_coherence_weight = 0.1;
  p->scaling_from_millimeters = 0.1;
  p->max_tanimoto_similarity = 0.9;
  p->conf = 0.5;
  p->proposal_engine_conf = 1.0;
  p->min_coverage = 0.5;
  p->min_triangle_area = 0.0;
  p->max_iters = 400;
  p->min_point_number = 6;
  p->max_model_number = 1;
  p->max_model_number_for_optimization = 5;
  p->use_prosac = 0;
  p->lo_iters = 8;
}

extern "C" int64_t epos_fit_workspace_bytes(int S, int64_t n_capacity,
                                            const EposFitParams* p, int32_t max_k) {
  if (!p || S < 0 || n_capacity < 0 || max_k < 1) return EPOS_E_INVALID;
  return make_layout(S, n_capacity, p->max_iters, max_k).total;
}

extern "C" int epos_find6d_poses_device(
    const double* xy, const double* xyz, const int64_t* slot_base, int S,
    int64_t n_capacity, const double* Ks, const int32_t* max_models,
    const uint64_t* seeds, const EposFitParams* p, int32_t max_k, void* work,
    double* poses, double* scores, int32_t* num_models, int32_t* labels,
    void* stream) {
  EPOS_REQUIRE(xy && xyz && slot_base && Ks && max_models && seeds && p && work &&
               poses && scores && num_models && labels, "null pointer");
  EPOS_REQUIRE(max_k >= 1 && p->max_iters >= 1, "max_k and max_iters must be >= 1");
  if (S == 0) return EPOS_OK;
  const Layout L = make_layout(S, n_capacity, p->max_iters, max_k);
  char* wb = static_cast<char*>(work);
  Work w;
  w.hyp_score = reinterpret_cast<double*>(wb + L.hyp_score);
  w.hyp_pose = reinterpret_cast<double*>(wb + L.hyp_pose);
  w.hyp_count = reinterpret_cast<int32_t*>(wb + L.hyp_count);
  w.active = reinterpret_cast<int32_t*>(wb + L.active);
  w.n_active = reinterpret_cast<int32_t*>(wb + L.n_active);
  w.done = reinterpret_cast<int32_t*>(wb + L.done);
  w.inl_bits = reinterpret_cast<uint64_t*>(wb + L.inl_bits);
  w.words_total = L.words_total;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(ransac_init, dim3(S), dim3(256), 0, st, slot_base, S, w, labels,
                     num_models, p->min_point_number, n_capacity);
  int rc = launch_status("ransac_init");
  if (rc) return rc;
  const dim3 hgrid(static_cast<unsigned>(ceil_div(p->max_iters, 4)), S);
  for (int round = 0; round < max_k; ++round) {
    hipLaunchKernelGGL(ransac_hypotheses, hgrid, dim3(256), 0, st, xy, xyz,
                       slot_base, Ks, seeds, max_models, num_models, *p, max_k,
                       round, w);
    rc = launch_status("ransac_hypotheses");
    if (rc) return rc;
    hipLaunchKernelGGL(ransac_select_refine, dim3(S), dim3(256), 0, st, xy, xyz,
                       slot_base, Ks, max_models, *p, max_k, w, poses, scores,
                       num_models, labels);
    rc = launch_status("ransac_select_refine");
    if (rc) return rc;
  }
  return EPOS_OK;
}

extern "C" int epos_find6d_poses(const double* xy, const double* xyz, int64_t n,
                                 const double* K, const EposFitParams* p,
                                 uint64_t seed, double* poses_out,
                                 int32_t* labels_out, double* scores_out,
                                 int32_t max_k) {
  EPOS_REQUIRE(K && p && poses_out && scores_out && (n == 0 || (xy && xyz && labels_out)),
               "null pointer");
  EPOS_REQUIRE(max_k >= 1, "max_k must be >= 1");
  for (int64_t i = 0; i < n; ++i) labels_out[i] = -1;
  if (n < p->min_point_number || n < 3) return 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    set_error("epos_find6d_poses: no HIP device (there is no CPU fallback)");
    return EPOS_E_NODEVICE;
  }
  int want = p->max_model_number;
  if (want < 0 || want > max_k) want = max_k;
  const int64_t wbytes = epos_fit_workspace_bytes(1, n, p, max_k);
  const int64_t sizes[] = {n * 16, n * 24, 16, 72, 8, 8, wbytes,
                           static_cast<int64_t>(max_k) * 96,
                           static_cast<int64_t>(max_k) * 8, 8, n * 4};
  char* d[11] = {0};
  int rc = EPOS_OK;
  for (int i = 0; i < 11 && !rc; ++i)
    rc = check_hip(hipMalloc(reinterpret_cast<void**>(&d[i]), sizes[i] + 8), "hipMalloc");
  int32_t k = 0;
  if (!rc) {
    const int64_t sb[2] = {0, n};
    const int32_t mm = want;
    rc = check_hip(hipMemcpy(d[0], xy, n * 16, hipMemcpyHostToDevice), "copy xy");
    if (!rc) rc = check_hip(hipMemcpy(d[1], xyz, n * 24, hipMemcpyHostToDevice), "copy xyz");
    if (!rc) rc = check_hip(hipMemcpy(d[2], sb, 16, hipMemcpyHostToDevice), "copy base");
    if (!rc) rc = check_hip(hipMemcpy(d[3], K, 72, hipMemcpyHostToDevice), "copy K");
    if (!rc) rc = check_hip(hipMemcpy(d[4], &mm, 4, hipMemcpyHostToDevice), "copy mm");
    if (!rc) rc = check_hip(hipMemcpy(d[5], &seed, 8, hipMemcpyHostToDevice), "copy seed");
    if (!rc)
      rc = epos_find6d_poses_device(
          reinterpret_cast<double*>(d[0]), reinterpret_cast<double*>(d[1]),
          reinterpret_cast<int64_t*>(d[2]), 1, n, reinterpret_cast<double*>(d[3]),
          reinterpret_cast<int32_t*>(d[4]), reinterpret_cast<uint64_t*>(d[5]), p,
          max_k, d[6], reinterpret_cast<double*>(d[7]),
          reinterpret_cast<double*>(d[8]), reinterpret_cast<int32_t*>(d[9]),
          reinterpret_cast<int32_t*>(d[10]), nullptr);
    if (!rc) rc = check_hip(hipDeviceSynchronize(), "sync");
    if (!rc) rc = check_hip(hipMemcpy(&k, d[9], 4, hipMemcpyDeviceToHost), "copy k");
    if (!rc && k > 0) {
      rc = check_hip(hipMemcpy(poses_out, d[7], static_cast<size_t>(k) * 96, hipMemcpyDeviceToHost), "copy poses");
      if (!rc) rc = check_hip(hipMemcpy(scores_out, d[8], static_cast<size_t>(k) * 8, hipMemcpyDeviceToHost), "copy scores");
    }
    if (!rc) rc = check_hip(hipMemcpy(labels_out, d[10], n * 4, hipMemcpyDeviceToHost), "copy labels");
  }
  for (int i = 0; i < 11; ++i)
    if (d[i]) (void)hipFree(d[i]);
  return rc ? rc : k;
}
