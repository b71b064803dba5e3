// Wavefront-parallel multi-instance PnP-RANSAC for gfx950 -- replaces the
// pyprogressivex.find6DPoses call of scripts/infer.py:470-488 (the un-vendored
// danini/progressive-x C++ module). Algorithm definition: DESIGN.md "Pose
// fitting" (P3P minimal solver, MSAC quality, Gauss-Newton local optimisation of
// the best hypothesis, spatial-coherence labelling, sequential multi-instance with
// Tanimoto / coverage tests, joint refinement).
//
// Mapping to the hardware (one call = one launch sequence over all slots = (image, object)
// pairs of a step; everything stays on the device, DESIGN.md (f) has the measurements):
//   * ransac_init: per-call state, the sweeps' position-ordered candidate stream (static
//     geometry) and the candidate window of every tile of 64 row-sorted points.
//   Per proposal round (max_k rounds, + 2 retries in a multi-instance search):
//   * ransac_hypotheses: ONE HYPOTHESIS SET PER WAVEFRONT. The 64 lanes compute the
//     (wave-uniform) P3P solve redundantly, then stride over the slot's correspondences; the
//     MSAC sums are reduced in a canonical order (64 strided partials, xor butterfly).
//     Grid = (max_iters / 4, slots).
//   * ransac_select_lo: FOUR workgroups per slot. Deterministic arg-max over the hypothesis
//     table, then the Gauss-Newton refits on the inliers: one pass over the points per refit
//     (score of the candidate + the 27 normal-equation sums of the step from it), 1024
//     strided partials in the oracle's canonical tree -- a reduce-scatter butterfly per wave,
//     a fence-free agent-scope hand-off between the workgroups, a pivot-free 6 x 6 solve by
//     every thread; last, the fixed-point residual table the labelling starts from.
//   * ransac_gc_scan (first sweep) / ransac_gc_delta (further sweeps): the spatial-coherence
//     labelling on the 5-D neighbourhood graph. Scan: sixteen waves per tile walk the tile's
//     window with the candidate in SCALAR registers (s_load), exact integer counters. Delta:
//     only the candidates the previous sweep flipped are revisited. (ransac_gc_sweep: the
//     LDS-staged predecessor, kept as EPOS_FIT_SCAN=0 and as ransac_nb_build, which writes
//     the neighbour lists the joint refinement walks.)
//   * ransac_refit_accept: refits on the labelled set (same machinery as select_lo), the
//     instance acceptance tests on inlier bitsets (ballot + popcount), stable in-place
//     compaction of the still-unexplained correspondences by the whole workgroup.
//   * pearl_*: joint refinement of multi-instance slots (labels over all instances).
// Arithmetic is fp64 (the reference hands f64 arrays to progressive-x) using only
// + - * / sqrt, compiled with -ffp-contract=off: results are reproducible run to
// run and identical to a scalar evaluation in the same canonical order.
#include <stdlib.h>

#include <algorithm>

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
  int32_t p[PF];
  __device__ __forceinline__ void load(const double* xy, const double* xyz,
                                       const int32_t* idx, int64_t i0, int stride,
                                       int64_t m) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int64_t i = i0 + static_cast<int64_t>(u) * stride;
      ok[u] = i < m;
      p[u] = idx ? idx[ok[u] ? i : i0] : static_cast<int32_t>(ok[u] ? i : i0);
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

#ifdef EPOS_FIT_TRACE   // tools/fit_trace.py: time stamps (100 MHz) of slot 0's first workgroup
__device__ unsigned long long g_fit_trace[8][32];
__device__ int g_fit_trace_n[8];
#define FIT_TRACE(K, COND)                                                      \
  do {                                                                          \
    if (COND) {                                                                 \
      const int i_ = g_fit_trace_n[K];                                          \
      if (i_ < 32) { g_fit_trace[K][i_] = wall_clock64(); g_fit_trace_n[K] = i_ + 1; } \
    }                                                                           \
  } while (0)
#else
#define FIT_TRACE(K, COND) do { } while (0)
#endif

// What a sweep needs of a candidate besides its static geometry. An inactive correspondence
// carries Z = +inf: its 5-D distance to anything is +inf, so it fails `d2 <= r2` without a
// label test.
struct __attribute__((aligned(16))) GcDyn {
  double Z;
  int32_t q;            // fixed-point residual
  int32_t out;          // 1: currently labelled outlier
};

// Raw sums of a full scan over a point's window, the point itself included: the number of
// neighbours, of neighbours labelled outlier, and the neighbours' fixed-point residuals.
struct __attribute__((aligned(16))) GcAcc {
  uint32_t deg, n0;
  unsigned long long S;
};

struct Work {
  double* hyp_score;     // [S][iters][4]
  double* hyp_pose;      // [S][iters][4][12]
  int32_t* hyp_count;    // [S][iters][4]
  int32_t* active;       // [N] local indices, pooled by slot_base
  int32_t* n_active;     // [S]
  int32_t* done;         // [S]
  uint64_t* inl_bits;    // [max_k][words_total]
  int64_t words_total;
  // proposal state between the kernels of one round
  double* cur_pose;      // [S][12]
  double* cur_score;     // [S]
  int32_t* cur_count;    // [S]
  int32_t* state;        // [S] 0: nothing to do, 1: label + refit, 2: refit / accept only
  int32_t* tries;        // [S] failed proposals since the last accepted instance
  int32_t* last_new;     // [S] newly explained points of the last accepted instance
  int32_t* gq;           // [N] fixed-point truncated residuals of the labelling
  uint8_t* lab_a;        // [N] labels, ping
  uint8_t* lab_b;        // [N] labels, pong
  uint8_t* lab_c;        // [N] joint refinement: the last sweep's labels in INDEX order
  int32_t* pearl_dt;     // [N][PEARL_DT] joint refinement: data terms of every point and label
  const int32_t* yorder; // [N] correspondences of a slot sorted by image y (or null:
  const int32_t* ypos;   //     they already are), and the inverse permutation
  // the sweeps' candidate stream, per POSITION of the row-sorted order (ransac_gc_scan):
  double* geo;           // [N][4] x, y, X, Y -- written once per call by ransac_init
  int32_t* win;          // [words_total][2] candidate window of every tile of 64 positions
  GcAcc* acc;            // [N] raw neighbour sums of the last full scan (self included)
  int8_t* flip_a;        // [N] label change of the last sweep per position: +1 became
  int8_t* flip_b;        //     outlier, -1 became inlier, 0 unchanged / not active (ping, pong)
  GcDyn* dyn_a;          // [N] Z (+inf: not active), residual, "labelled outlier": ping
  GcDyn* dyn_b;          //     pong (a sweep reads one and writes the other)
  // neighbour lists of the 5-D graph (built once per call: the graph depends on neither
  // labels nor rounds), per POSITION of the row-sorted order
  uint16_t* nb_cnt;      // [N][GC_W] entries of the point's GC_W sub-lists
  int16_t* nb_pool;      // [N][GC_W][NB_SUB] position deltas
  int32_t* nb_ok;        // [S] 1: lists complete; 0: overflow -> the sweeps scan windows
  // cooperative local optimisation (LO_G workgroups per slot)
  unsigned* lo_cnt;      // [S][lo_launches(max_k)] arrival counters, zeroed by ransac_init
  double* lo_data;       // [S][2][LO_G * 4][LO_NV] wave sums, double buffered
  int32_t* lo_timeout;   // [1] sticky: a workgroup gave up waiting for its siblings
  unsigned lo_spin_max;  // polls before that happens (LO_SPIN_MAX; EPOS_FIT_SPIN_MAX: tests)
  // joint refinement
  double* pearl_pose;            // [S][8][12] candidate poses
  unsigned long long* pearl_acc; // [S][4][PEARL_BINS] data / smoothness sums before, after (exact)
  int32_t* pearl_state;          // [S] 1: refining
  int32_t* pearl_moved;          // [S]
};

constexpr int GC_Q = 1 << 20;          // fixed point of the labelling energies
constexpr int PEARL_DT = 9;            // = PEARL_MAX_K + 1 columns of Work::pearl_dt
constexpr double LO_MIN_GAIN = 1e-6;   // relative MSAC gain below which the refits stop

// b^e by binary exponentiation (multiplications only: every implementation of the
// stage gets the same bits)
__device__ __forceinline__ double powi(double b, int64_t e) {
  double r = 1.0;
  while (e > 0) {
    if (e & 1) r = r * b;
    b = b * b;
    e >>= 1;
  }
  return r;
}

// A proposal round of slot s failed (no model, too few inliers, Tanimoto / coverage):
// single-instance search stops; the multi-instance search retries with fresh samples
// while the samples drawn since the last success have not reached confidence `conf` of
// having hit an instance as large as the last accepted one (DESIGN.md "Pose fitting").
__device__ void round_failed(int s, const Work& w, const EposFitParams& prm, int want,
                             int k, int64_t n_active) {
  const int tries = ++w.tries[s];
  bool stop = want == 1;
  if (!stop) {
    const int64_t ref = k == 0 ? prm.min_point_number : w.last_new[s];
    if (ref >= n_active) {
      stop = true;
    } else {
      const double r = static_cast<double>(ref) / static_cast<double>(n_active);
      stop = !(powi(1.0 - r * r * r, static_cast<int64_t>(tries) * prm.max_iters) >
               1.0 - prm.conf);
    }
  }
  if (stop) w.done[s] = 1;
}

__global__ __launch_bounds__(256) void ransac_init(const double* __restrict__ xy_all,
                                                   const double* __restrict__ xyz_all,
                                                   const int64_t* slot_base, int S,
                                                   Work w, int32_t* labels,
                                                   int32_t* num_models,
                                                   int min_pts, int64_t n_capacity, int n_lo,
                                                   int build_nb, int build_geo, double rad) {
  const int s = blockIdx.x;
  const bool first = blockIdx.y == 0;        // gridDim.y workgroups share the slot's rows
  if (first) {
    for (int i = threadIdx.x; i < n_lo; i += blockDim.x)
      w.lo_cnt[static_cast<int64_t>(s) * n_lo + i] = 0u;
    if (threadIdx.x == 0) w.nb_ok[s] = build_nb;
    if (s == 0 && threadIdx.x == 0) *w.lo_timeout = 0;
  }
  const int64_t base = slot_base[s];
  // A slot whose rows would end beyond the pooled arrays (the correspondence stage
  // raised its overflow flag and wrote nothing there) is fitted as EMPTY: no kernel of
  // this stage then touches a row >= n_capacity. The host reports the overflow.
  const int64_t n = slot_base[s + 1] <= n_capacity ? slot_base[s + 1] - base : 0;
  for (int64_t i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += blockDim.x * gridDim.y) {
    w.active[base + i] = static_cast<int32_t>(i);
    labels[base + i] = -1;
    if (build_geo) {                  // static geometry of the sweeps' candidate stream
      const int64_t o = w.yorder ? w.yorder[base + i] : i;
      double* g = w.geo + 4 * (base + i);
      g[0] = xy_all[2 * (base + o)]; g[1] = xy_all[2 * (base + o) + 1];
      g[2] = xyz_all[3 * (base + o)]; g[3] = xyz_all[3 * (base + o) + 1];
    }
  }
  if (build_geo) {
    // The candidate window of every tile of 64 positions -- the positions whose image row
    // can hold a neighbour of a point of the tile -- depends on the geometry only: found
    // here once per call (one wavefront per tile, 64-ary searches over the sorted rows: three
    // dependent loads per side) instead of by every sweep.
    const int lane = threadIdx.x & 63;
    const int64_t wave = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t waves = static_cast<int64_t>(gridDim.y) * (blockDim.x >> 6);
    const int32_t* yo = w.yorder ? w.yorder + base : nullptr;
    const double* ys = xy_all + 2 * base + 1;
    int32_t* win = w.win + 2 * (base / 64 + s);
    for (int64_t tile = wave; tile * 64 < n; tile += waves) {
      const int64_t pos0 = tile * 64;
      const int64_t last = pos0 + 63 < n ? pos0 + 63 : n - 1;
      const double ylo = ys[2 * (yo ? yo[pos0] : pos0)] - rad;
      const double yhi = ys[2 * (yo ? yo[last] : last)] + rad;
      for (int side = 0; side < 2; ++side) {
        // side 0: first position in [0, pos0] with y >= ylo; side 1: first position in
        // [last + 1, n] with y > yhi   (y is non-decreasing along the sorted order)
        int64_t lo = side ? last + 1 : 0, hi = side ? n : pos0;
        while (hi - lo > 0) {
          const int64_t step = (hi - lo + 63) / 64;
          const int64_t q = lo + lane * step;
          bool pred = false;
          if (q < hi) {
            const double yq = ys[2 * (yo ? yo[q] : q)];
            pred = side ? yq > yhi : yq >= ylo;
          }
          const unsigned long long b = __ballot(pred);
          if (!b) {                 // no probe at or beyond the bound: it lies after the last
            const int64_t tv = (hi - lo - 1) / step;
            lo = lo + tv * step + 1;
          } else {                  // the bound lies in (probe[first - 1], probe[first]]
            const int f = __ffsll(static_cast<long long>(b)) - 1;
            const int64_t qf = lo + static_cast<int64_t>(f) * step;
            lo = f == 0 ? lo : lo + static_cast<int64_t>(f - 1) * step + 1;
            hi = qf;
          }
        }
        if (lane == 0) win[2 * tile + side] = static_cast<int32_t>(lo);
      }
    }
  }
  if (first && threadIdx.x == 0) {
    w.n_active[s] = static_cast<int32_t>(n);
    num_models[s] = 0;
    w.done[s] = (n < min_pts || n < 3) ? 1 : 0;      // infer.py:420-422
    w.state[s] = 0;
    w.tries[s] = 0;
    w.last_new[s] = min_pts;
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
    const int64_t n = slot_base[s + 1] - base;
    double K[9];
    for (int i = 0; i < 9; ++i) K[i] = Ks[s * 9 + i];
    int64_t m = n_active;
    if (prm.use_prosac) {
      m = (n_active * static_cast<int64_t>(it + 1) + prm.max_iters - 1) / prm.max_iters;
      if (m < prm.min_point_number) m = prm.min_point_number;
      if (m < 3) m = 3;
      if (m > n_active) m = n_active;
    }
    FIT_TRACE(0, s == 0 && it == 0 && lane == 0);
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
      FIT_TRACE(0, s == 0 && it == 0 && lane == 0);
      ns = p3p(f, X, sols);
      FIT_TRACE(0, s == 0 && it == 0 && lane == 0);
      const double thr2 = prm.threshold * prm.threshold;
      double* hp = w.hyp_pose + (static_cast<int64_t>(s) * prm.max_iters + it) * MAX_SOL * 12;
      int32_t* hc = w.hyp_count + (static_cast<int64_t>(s) * prm.max_iters + it) * MAX_SOL;
      double sc[MAX_SOL];
      int cnt[MAX_SOL];
      // while nothing has been removed the active list is the identity (ransac_init): the
      // scoring passes then index the points directly -- one dependent load per item less
      score_poses(sols, ns, K, xy, xyz, n_active == n ? nullptr : active, n_active, thr2, lane,
                  sc, cnt);
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
  FIT_TRACE(0, s == 0 && it == 0 && lane == 0);
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

// Gaussian elimination WITHOUT pivoting (see below), the arithmetic of the C definition's
// solve6 operation for operation, with every index a compile-time constant so that the
// 6 x 7 system stays in registers. The indexed form went through scratch memory -- a
// dependent chain of ~300 private loads and stores that every thread of the workgroup
// repeated in every refit pass (~6 us of each ~17 us pass).
__device__ int solve6(const double* H /*[36]*/, const double* g, double* x) {
  // H = J^T J is symmetric positive (semi)definite: elimination WITHOUT pivoting is stable
  // for it, and one reciprocal per pivot serves the column and the back-substitution (round
  // 3: the pivot search + row swaps through selects and 21 fp64 divisions were 2.5 us of
  // every 10 us refit pass; the oracle's solve6 is the same arithmetic)
  double A[6][7], inv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) A[i][j] = H[i * 6 + j];
    A[i][6] = -g[i];
  }
  bool bad = false;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    bad = bad || !(A[c][c] > 1e-300);
    inv[c] = 1.0 / A[c][c];
#pragma unroll
    for (int r = c + 1; r < 6; ++r) {
      const double fct = A[r][c] * inv[c];
#pragma unroll
      for (int j = c + 1; j < 7; ++j) A[r][j] -= fct * A[c][j];
    }
  }
  if (bad) return 1;
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double sacc = A[i][6];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) sacc -= A[i][j] * x[j];
    x[i] = sacc * inv[i];
  }
  return 0;
}

// ---- sums of the local optimisation ------------------------------------------------------
// Canonical order over P = 256 * G strided partials (partial q takes items q, q + P, ...):
// the xor butterfly inside each group of 64, (w0 + w1) + (w2 + w3) inside each group of 256,
// then (W0 + W1) + (W2 + W3) over the G = 4 groups of 256 (G = 1: the group of 256 alone).
// G = 1 is one 256-thread workgroup (joint refinement). G = LO_G = 4 (round 3): FOUR
// workgroups per slot, workgroup g owning partials [256 g, 256 g + 256): each pass of a refit
// is a latency-bound sweep over the slot's correspondences (a dependent gather and ~250 fp64
// operations per item), so four workgroups on four CUs shorten it almost four-fold; the
// sixteen wave sums meet in global memory (placement-independent agent-scope hand-off of
// cdna_hip_programming.md Guideline 16: plain stores, every wave drains, ONE release + ticket
// per workgroup, relaxed polling, ONE acquire) and every workgroup combines them in the same
// order, so all of them continue with identical values and identical control flow. The
// siblings of a slot have adjacent block indices, i.e. the same position in their XCDs'
// dispatch order: whenever one of them is resident the others are or are about to be; the
// spin is bounded anyway (lo_timeout).
constexpr int LO_G = 4;
constexpr int LO_NV = 32;            // doubles per wave row (27 sums, score, count)
constexpr unsigned LO_SPIN_MAX = 1u << 24;

struct LoSync {
  int trace = 0;        // EPOS_FIT_TRACE: this workgroup stamps the phases of its passes
  unsigned* cnt;        // this launch's counter of the slot (null: single workgroup)
  double* data;         // [2][LO_G * 4][LO_NV]
  int32_t* timeout;
  unsigned spin_max;    // polls before a workgroup gives up on its siblings (Work::lo_spin_max)
  int g;                // workgroup index within the slot
  unsigned epoch;       // exchanges done so far in this launch
};

// The wave sums of up to 32 quantities at once. A butterfly per quantity (butterfly_sum:
// xor 32, 16, ... 1, every lane ends with the total) moves 6 x 2 words per quantity; here the
// first five levels HALVE the set instead -- at the level of xor-distance d a lane keeps the
// quantities whose next index bit equals its own bit d and receives its partner's copies of
// exactly those -- so 16 + 8 + 4 + 2 + 1 + 1 values cross lanes instead of 6 x 32. Every
// quantity is still summed over the same pairs in the same order (a + b == b + a bit for
// bit), i.e. the canonical tree of the oracle's tree64. Lane l ends with the total of
// quantity lo_owned(l).
__device__ __forceinline__ int lo_owned(int lane) {        // bits 5..1 of the lane, reversed
  return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) |
         (((lane >> 2) & 1) << 3) | (((lane >> 1) & 1) << 4);
}
__device__ __forceinline__ double reduce_scatter32(double (&a)[32], int lane) {
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int off = 32 >> k;
    const bool up = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < (16 >> k); ++i) {
      const double keep = up ? a[2 * i + 1] : a[2 * i];
      const double send = up ? a[2 * i] : a[2 * i + 1];
      a[i] = keep + __shfl_xor(send, off, 64);
    }
  }
  return a[0] + __shfl_xor(a[0], 1, 64);
}

// val: the total over this WAVE of quantity lo_owned(lane) (reduce_scatter32). Returns the
// canonical totals in comb[0..nv) (LDS, valid for every thread after the call). s_rows:
// LDS [4][LO_NV].
__device__ void lo_combine(LoSync& sy, double val, int nv, int t, double* s_rows,
                           double* comb) {
  const int lane = t & 63, wave = t >> 6;
  const int own = lo_owned(lane);
  const bool writer = (lane & 1) == 0 && own < nv;
  if (sy.cnt == nullptr) {                       // one workgroup: through LDS only
    if (writer) s_rows[wave * LO_NV + own] = val;
    __syncthreads();
    if (t < nv)
      comb[t] = (s_rows[t] + s_rows[LO_NV + t]) + (s_rows[2 * LO_NV + t] + s_rows[3 * LO_NV + t]);
    __syncthreads();
    return;
  }
  // The exchanged words are written and read by AGENT-SCOPE ATOMIC stores / loads (sc1: they
  // are coherent across the XCDs' L2s by themselves), so the hand-off needs no release /
  // acquire fence -- on this part a fence is a write-back / invalidate of the whole L2,
  // which costs microseconds while other streams' GEMMs keep it dirty: every storing wave
  // waits for its stores to complete, the workgroup meets, ONE relaxed ticket, relaxed
  // polling, and the reads are issued after the poll returned (they depend on it).
  unsigned long long* buf = reinterpret_cast<unsigned long long*>(sy.data) +
                            (sy.epoch & 1u) * (LO_G * 4 * LO_NV);
  if (writer)
    __hip_atomic_store(buf + (sy.g * 4 + wave) * LO_NV + own,
                       __builtin_bit_cast(unsigned long long, val), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
  __syncthreads();
  if (t == 0) {
    __hip_atomic_fetch_add(sy.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = LO_G * (sy.epoch + 1u);
    unsigned spins = 0;
    while (__hip_atomic_load(sy.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      // a hand-off of this launch has already timed out (the call will report
      // num_models = -1): do not spin the full budget again at every later exchange
      if ((spins & 255u) == 255u &&
          __hip_atomic_load(sy.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        break;
      if (++spins > sy.spin_max) {
        __hip_atomic_store(sy.timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (t < nv) {
    double W[LO_G];
#pragma unroll
    for (int g = 0; g < LO_G; ++g) {
      const unsigned long long* r = buf + (g * 4) * LO_NV + t;
      double r4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        r4[q] = __builtin_bit_cast(double, __hip_atomic_load(r + q * LO_NV, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
      W[g] = (r4[0] + r4[1]) + (r4[2] + r4[3]);
    }
    comb[t] = (W[0] + W[1]) + (W[2] + W[3]);
  }
  __syncthreads();
  ++sy.epoch;
}

struct LoLds {                 // LDS of a local-optimisation workgroup
  double rows[4 * LO_NV];
  double comb[LO_NV];
};

// One pass over the correspondences idx[0..m) (idx == nullptr: 0..m) that yields
//   * SCORE: the MSAC score and inlier count of `pose` at thr2,
//   * the Gauss-Newton step from `pose` on its members -- lab == nullptr: the inliers at
//     thr2; otherwise the points with lab[p] == sel (fixed membership, full weight) --
//     as the next pose (returns 1 when the normal equations are singular / non-finite).
// One pass per refit: the step of an accepted candidate is already there when the next
// refit starts. All sums in the canonical order of lo_combine (G = 1 or LO_G workgroups).
template <bool SCORE>
__device__ int lo_pass(const double* pose, const double* K, const double* xy,
                       const double* xyz, const int32_t* idx, int64_t m, double thr2, int t,
                       LoSync& sy, LoLds* lds, const uint8_t* lab, int sel, double* score,
                       int* count, double* next) {
  const int G = sy.cnt ? LO_G : 1;
  const int stride = 256 * G;
  const double inv_thr2 = 1.0 / thr2;
  double acc[32];
#pragma unroll
  for (int v = 0; v < 32; ++v) acc[v] = 0.0;
  int cnt = 0;
  FIT_TRACE(4, sy.trace && t == 0);
  for (int64_t i0 = static_cast<int64_t>(sy.g) * 256 + t; i0 < m;
       i0 += static_cast<int64_t>(stride) * PF) {
    PointBatch pb;
    pb.load(xy, xyz, idx, i0, stride, m);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (!pb.ok[u]) continue;
      double e2, Xc[3], r[2];
      if (reproj(pose, K, pb.x2[u], pb.x3[u], &e2, Xc, r)) continue;
      const bool inl = e2 < thr2;
      if (SCORE && inl) { acc[27] += 1.0 - e2 * inv_thr2; ++cnt; }
      if (lab ? lab[pb.p[u]] != sel : !inl) continue;
      const double iz = 1.0 / Xc[2];
      const double a0 = K[0] * iz, a1 = K[1] * iz,
                   a2 = -(K[0] * Xc[0] + K[1] * Xc[1]) * iz * iz;
      const double b1 = K[4] * iz, b2 = -(K[4] * Xc[1]) * iz * iz;
      double J0[6], J1[6];
      // Xc(w) = Xc - [Xc]x w: row . (-[Xc]x)  (sign fixed in round 2)
      J0[0] = -a1 * Xc[2] + a2 * Xc[1];
      J0[1] = a0 * Xc[2] - a2 * Xc[0];
      J0[2] = -a0 * Xc[1] + a1 * Xc[0];
      J0[3] = a0; J0[4] = a1; J0[5] = a2;
      J1[0] = -b1 * Xc[2] + b2 * Xc[1];
      J1[1] = -b2 * Xc[0];
      J1[2] = b1 * Xc[0];
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
  constexpr int NV = SCORE ? 29 : 27;
  FIT_TRACE(4, sy.trace && t == 0);
  if (SCORE) acc[28] = static_cast<double>(cnt);     // exact in any order: counts < 2^31
  const double mine = reduce_scatter32(acc, t & 63);
  FIT_TRACE(4, sy.trace && t == 0);
  lo_combine(sy, mine, NV, t, lds->rows, lds->comb);
  FIT_TRACE(4, sy.trace && t == 0);
  const double* tot = lds->comb;
  if (SCORE) {
    *score = tot[27];
    *count = static_cast<int>(tot[28]);
  }
  double H[36], g[6], x[6];
  {
    int v = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 6; ++b) { H[a * 6 + b] = tot[v]; H[b * 6 + a] = tot[v]; ++v; }
#pragma unroll
    for (int a = 0; a < 6; ++a) g[a] = tot[v++];
  }
  __syncthreads();               // every thread has read the totals: comb may be reused
  if (solve6(H, g, x)) return 1;
  double qw = 1.0, qx = 0.5 * x[0], qy = 0.5 * x[1], qz = 0.5 * x[2];
  const double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  qw /= qn; qx /= qn; qy /= qn; qz /= qn;
  double dR[9];
  dR[0] = 1.0 - 2.0 * (qy * qy + qz * qz); dR[1] = 2.0 * (qx * qy - qz * qw); dR[2] = 2.0 * (qx * qz + qy * qw);
  dR[3] = 2.0 * (qx * qy + qz * qw); dR[4] = 1.0 - 2.0 * (qx * qx + qz * qz); dR[5] = 2.0 * (qy * qz - qx * qw);
  dR[6] = 2.0 * (qx * qz - qy * qw); dR[7] = 2.0 * (qy * qz + qx * qw); dR[8] = 1.0 - 2.0 * (qx * qx + qy * qy);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      next[i * 3 + j] = dR[i * 3] * pose[j] + dR[i * 3 + 1] * pose[3 + j] + dR[i * 3 + 2] * pose[6 + j];
    next[9 + i] = dR[i * 3] * pose[9] + dR[i * 3 + 1] * pose[10] + dR[i * 3 + 2] * pose[11] + x[3 + i];
  }
  bool bad = false;
#pragma unroll
  for (int i = 0; i < 12; ++i) bad = bad || !(next[i] == next[i]);
  FIT_TRACE(4, sy.trace && t == 0);
  return bad ? 1 : 0;
}

// ---- round, step 2: best hypothesis (with the RANSAC confidence bound) + local
// optimisation (i) + the residual table and thresholded labels of the labelling step.
__global__ __launch_bounds__(256) void ransac_select_lo(
    const double* __restrict__ xy_all, const double* __restrict__ xyz_all,
    const int64_t* __restrict__ slot_base, const double* __restrict__ Ks,
    const int32_t* __restrict__ max_models, EposFitParams prm, int max_k, int round,
    Work w, const int32_t* __restrict__ num_models, const int32_t* __restrict__ labels_all,
    int lo_launch, int n_lo) {
  __shared__ double s_score[256];
  __shared__ int s_index[256];
  __shared__ LoLds s_lo;
  // LO_G workgroups per slot (blockIdx.x = g: siblings are adjacent in dispatch order);
  // all of them take every decision from the same data, workgroup 0 writes the state
  const int s = blockIdx.y;
  const int g = blockIdx.x;
  const int t = threadIdx.x;
  if (t == 0 && g == 0) w.state[s] = 0;
  FIT_TRACE(1, s == 0 && g == 0 && t == 0);
  if (w.done[s]) return;                                   // uniform over the slot
  int want = max_models[s];
  if (want < 0 || want > max_k) want = max_k;
  const int k = num_models[s];
  const int64_t base = slot_base[s];
  const int64_t n = slot_base[s + 1] - base;
  int32_t* active = w.active + base;
  const int64_t n_active = w.n_active[s];
  if (k >= want || round >= want + (want > 1 ? 2 : 0) ||
      n_active < prm.min_point_number || n_active < 3) {
    if (t == 0 && g == 0) w.done[s] = 1;
    return;
  }
  const int nh = prm.max_iters * MAX_SOL;
  const double* hs = w.hyp_score + static_cast<int64_t>(s) * nh;
  const int32_t* hcnt = w.hyp_count + static_cast<int64_t>(s) * nh;
  if (prm.proposal_engine_conf < 1.0) {
    // sequential semantics of the termination bound: the best among the hypotheses
    // drawn before (1 - w^3)^it <= 1 - conf (sequential semantics)
    if (t == 0) {
      double best = -1.0;
      int best_i = 0x7fffffff, best_c = 0;
      for (int it = 0; it < prm.max_iters; ++it) {
        if (it > 0 && best > 0.0) {
          const double r = static_cast<double>(best_c) / static_cast<double>(n_active);
          if (powi(1.0 - r * r * r, it) <= 1.0 - prm.proposal_engine_conf) break;
        }
        for (int q = 0; q < MAX_SOL; ++q) {
          const double v = hs[it * MAX_SOL + q];
          if (v > best) { best = v; best_i = it * MAX_SOL + q; best_c = hcnt[best_i]; }
        }
      }
      s_score[0] = best; s_index[0] = best_i;
    }
    __syncthreads();
  } else {
    // ---- arg-max over the hypothesis table (ties -> lowest index) ----
    double best = -1.0;
    int best_i = 0x7fffffff;
    for (int i = t; i < nh; i += 256) {
      const double v = hs[i];
      if (v > best) { best = v; best_i = i; }
    }
    // (max score, lowest index) is associative and commutative: any reduction order gives
    // the same winner -- a butterfly per wave, then the four wave winners through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double v = __shfl_xor(best, off, 64);
      const int vi = __shfl_xor(best_i, off, 64);
      if (v > best || (v == best && vi < best_i)) { best = v; best_i = vi; }
    }
    if ((t & 63) == 0) { s_score[t >> 6] = best; s_index[t >> 6] = best_i; }
    __syncthreads();
    best = s_score[0]; best_i = s_index[0];
#pragma unroll
    for (int g2 = 1; g2 < 4; ++g2) {
      const double v = s_score[g2];
      const int vi = s_index[g2];
      if (v > best || (v == best && vi < best_i)) { best = v; best_i = vi; }
    }
    __syncthreads();
    if (t == 0) { s_score[0] = best; s_index[0] = best_i; }
    __syncthreads();
  }
  double best_score = s_score[0];
  const int bi = s_index[0];
  int best_count = best_score > 0.0 ? hcnt[bi] : 0;
  FIT_TRACE(1, s == 0 && g == 0 && t == 0);
  if (!(best_score > 0.0) || best_count < 3) {             // uniform
    if (t == 0 && g == 0) round_failed(s, w, prm, want, k, n_active);
    return;
  }
  LoSync sy;
  sy.cnt = w.lo_cnt + static_cast<int64_t>(s) * n_lo + lo_launch;
  sy.data = w.lo_data + static_cast<int64_t>(s) * (2 * LO_G * 4 * LO_NV);
  sy.timeout = w.lo_timeout;
  sy.spin_max = w.lo_spin_max;
  sy.g = g;
  sy.epoch = 0;
#ifdef EPOS_FIT_TRACE
  sy.trace = s == 0 && g == 0;
#endif
  const double* xy = xy_all + 2 * base;
  const double* xyz = xyz_all + 3 * base;
  double K[9];
  for (int i = 0; i < 9; ++i) K[i] = Ks[s * 9 + i];
  const double thr2 = prm.threshold * prm.threshold;
  double pose[12];
  for (int i = 0; i < 12; ++i) pose[i] = w.hyp_pose[(static_cast<int64_t>(s) * nh + bi) * 12 + i];
  // ---- local optimisation (i): the whole workgroup sums, every thread steps ----
  orthonormalize(pose);
  // each pass scores a pose AND takes the Gauss-Newton step from it: the step of an
  // accepted candidate is already there when the next refit starts
  double cand[12];
  const int32_t* idx = n_active == n ? nullptr : active;   // identity while nothing is removed
  FIT_TRACE(1, s == 0 && g == 0 && t == 0);
  int fail = lo_pass<true>(pose, K, xy, xyz, idx, n_active, thr2, t, sy, &s_lo, nullptr, 1,
                           &best_score, &best_count, cand);
  FIT_TRACE(1, s == 0 && g == 0 && t == 0);
  for (int li = 0; li < prm.lo_iters && !fail; ++li) {
    double sc, cand2[12];
    int cnt;
    const int fail2 = lo_pass<true>(cand, K, xy, xyz, idx, n_active, thr2, t, sy, &s_lo,
                                    nullptr, 1, &sc, &cnt, cand2);
    FIT_TRACE(1, s == 0 && g == 0 && t == 0);
    if (!(sc > best_score)) break;
    const double gain = sc - best_score;
    best_score = sc; best_count = cnt;
    for (int i = 0; i < 12; ++i) { pose[i] = cand[i]; cand[i] = cand2[i]; }
    fail = fail2;
    if (!(gain > LO_MIN_GAIN * sc)) break;        // converged: further steps are noise
  }
  if (g == 0) {
    if (t < 12) w.cur_pose[s * 12 + t] = pose[t];
    if (t == 0) { w.cur_score[s] = best_score; w.cur_count[s] = best_count; }
  }
  const bool gc = prm.gc_sweeps > 0 && prm.spatial_coherence_weight > 0.0 &&
                  prm.neighborhood_ball_radius > 0.0;
  if (!gc) { if (t == 0 && g == 0) w.state[s] = 2; return; }
  // ---- residual table (2^-20 fixed point of min(e^2 / (1.5 tau)^2, 1)) and the
  //      thresholded labelling the sweeps start from; 2 = not active. A correspondence is
  //      active iff no accepted instance explains it yet (labels < 0: ransac_refit_accept
  //      labels exactly the ones it removes from the active list), so every point is
  //      decided by the one thread that owns it -- the workgroups of the slot share the rows
  uint8_t* lab = w.lab_a + base;
  int32_t* gq = w.gq + base;
  const int32_t* labels = labels_all + base;
  const double tthr = 1.5 * prm.threshold, tthr2 = tthr * tthr;
  GcDyn* dyn = w.dyn_a + base;
  const int32_t* ypos = w.ypos ? w.ypos + base : nullptr;
  for (int64_t p = static_cast<int64_t>(g) * 256 + t; p < n; p += 256 * LO_G) {
    const int64_t pos = ypos ? ypos[p] : p;
    GcDyn dn;
    dn.Z = xyz[3 * p + 2]; dn.q = GC_Q; dn.out = 1;
    if (labels[p] >= 0) {
      lab[p] = 2;
      dn.Z = __builtin_huge_val(); dn.q = 0; dn.out = 0;
      dyn[pos] = dn;
      continue;
    }
    double e2, Xc[3], r[2];
    if (reproj(pose, K, xy + 2 * p, xyz + 3 * p, &e2, Xc, r)) {
      gq[p] = GC_Q; lab[p] = 0; dyn[pos] = dn;
      continue;
    }
    double d = e2 / tthr2;
    if (!(d < 1.0)) d = 1.0;
    dn.q = static_cast<int32_t>(d * static_cast<double>(GC_Q));
    dn.out = e2 < thr2 ? 0 : 1;
    gq[p] = dn.q;
    lab[p] = e2 < thr2 ? 1 : 0;
    dyn[pos] = dn;
  }
  if (t == 0 && g == 0) w.state[s] = 1;
  FIT_TRACE(1, s == 0 && g == 0 && t == 0);
}

// ---- round, step 3 (gc_sweeps launches): one synchronous relabelling sweep of the
// spatial-coherence energy (GC-RANSAC labelling). ONE WAVEFRONT PER POINT: the
// lanes walk the point's window of the y-sorted correspondences (two correspondences
// can only be neighbours when their image rows are within tau_d), test the 5-D
// distance and count degree / residual sum / outlier-labelled neighbours in integers
// (exact, order independent); lane 0 takes the cheaper label.
__device__ __forceinline__ bool gc_neighbours(const double* xy, const double* xyz,
                                              int32_t a, int32_t b, double s2, double r2) {
  const double dx = xy[2 * a] - xy[2 * b], dy = xy[2 * a + 1] - xy[2 * b + 1];
  const double dX = xyz[3 * a] - xyz[3 * b], dY = xyz[3 * a + 1] - xyz[3 * b + 1],
               dZ = xyz[3 * a + 2] - xyz[3 * b + 2];
  const double d2 = (dx * dx + dy * dy) + s2 * ((dX * dX + dY * dY) + dZ * dZ);
  return d2 <= r2;
}

// a slot's neighbour lists (null cnt: not available, scan the window)
constexpr int NB_W = 4;            // sub-lists per point (= waves of a sweep workgroup)
constexpr int NB_SUB = 160;        // entries per sub-list: 640 neighbours per point
struct NbLists {
  const uint16_t* cnt;
  const int16_t* pool;
};

__device__ __forceinline__ int64_t butterfly_sum_i64(int64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Visits every ACTIVE neighbour of point p (slot-local index): f(q, valid), q = the
// neighbour's POSITION in the row-sorted order (its index if the slot is not sorted) -- f is
// called by every lane of the wave, `valid` says whether the lane holds a neighbour (q is a
// safe position otherwise), so that f's own loads are unconditional and can be in flight
// together. Callers keep what they look up per neighbour in position order: a neighbour is
// then one list entry and one nearby byte, no order look-up. Wave-wide.
// LISTS: the slot's neighbour lists are complete (Work::nb_ok) -- the caller's kernel is
// instantiated once per case, because the window walk of the other case (fp64 distance
// tests) costs the list walk three of eight waves per SIMD if both sit in one kernel.
template <bool LISTS, typename F>
__device__ __forceinline__ void for_each_neighbour(const double* xy, const double* xyz,
                                                   int64_t n, int32_t p,
                                                   const int32_t* yorder, const int32_t* ypos,
                                                   double rad, double s2, double r2, int lane,
                                                   const NbLists& nl, F f) {
  const int64_t pos = ypos ? ypos[p] : p;
  if constexpr (LISTS) {     // the slot's neighbour lists are complete: walk the point's
    // One quarter-wave per sub-list, two entries per lane and trip: count -> list entries ->
    // f's loads = THREE dependent round trips for up to 128 neighbours (round 4; one
    // sub-list after the other with 64 lanes each, and an order look-up per neighbour, was
    // 3 per sub-list + 1 = 13).
    static_assert(NB_W == 4, "one quarter-wave per sub-list");
    const int q = lane >> 4, l16 = lane & 15;
    const int cntq = nl.cnt[pos * NB_W + q];
    const int16_t* lst = nl.pool + (pos * NB_W + q) * NB_SUB;
    int cmax = cntq;                                   // wave-uniform trip count
    cmax = max(cmax, __shfl_xor(cmax, 16, 64));
    cmax = max(cmax, __shfl_xor(cmax, 32, 64));
    for (int i0 = 0; i0 < cmax; i0 += 32) {
      const int ia = i0 + l16, ib = ia + 16;
      const bool va = ia < cntq, vb = ib < cntq;
      const int64_t pa = pos + (va ? lst[ia] : 0), pb = pos + (vb ? lst[ib] : 0);
      f(pa, va);
      f(pb, vb);
    }
    return;
  }
  const double yp = xy[2 * p + 1];
  for (int dir = 0; dir < 2; ++dir) {
    for (int64_t j0 = dir ? pos + 1 : pos - 1; dir ? j0 < n : j0 >= 0; j0 += dir ? 64 : -64) {
      const int64_t j = dir ? j0 + lane : j0 - lane;
      const bool in = dir ? j < n : j >= 0;
      int32_t o = 0;
      bool near = false;
      if (in) {
        o = yorder ? yorder[j] : static_cast<int32_t>(j);
        near = !(fabs(yp - xy[2 * o + 1]) > rad);
      }
      if (!__any(near)) break;            // sorted by y: nothing further can be in range
      f(in ? j : pos, near && gc_neighbours(xy, xyz, p, o, s2, r2));
    }
  }
}

// One workgroup per TILE of 64 consecutive points of the row-sorted order. The points of
// a tile share one candidate window (all correspondences whose image row is within tau_d
// of the tile's rows: a contiguous range of the sorted order, found by two binary
// searches); the window streams through LDS in chunks of GC_T candidates and each thread
// tests its point (t & 63) against its wave's share of the chunk (t >> 6) -- all 64 lanes of a
// wave read the SAME candidate: LDS broadcasts, no bank conflicts. Per pair: the 5-D
// distance in fp64 and three integer counters (exact, order independent). The first
// version (one wavefront per point walking its own window through global memory) cost
// 0.15-0.6 ms per sweep: a chain of dependent gathers per point and 6 global loads per
// candidate pair.
// 256 threads per tile: 1024 (16 waves share a tile's window) is 20 % faster for one
// image at a time (fitting 0.55 vs 0.66 ms) but costs throughput with several images in
// flight (310 vs 320 images/s, same box: the large workgroups displace GEMM workgroups)
#ifndef EPOS_GS_WAVES
#define EPOS_GS_WAVES 16
#endif
#ifndef EPOS_GS_POINTS
#define EPOS_GS_POINTS 1
#endif
#ifndef EPOS_GC_THREADS
#define EPOS_GC_THREADS 256
#endif
#ifdef EPOS_GC_STATS              // tools/gc_stats.py: tiles visited, candidates streamed
__device__ unsigned long long g_gc_stats[4];
#endif
struct GcCand {
  double x, y, X, Y, Z;
  int32_t q;          // fixed-point residual
  int32_t ol;         // slot-local index | label << 30
};
constexpr int GC_T = EPOS_GC_THREADS;  // threads per tile workgroup
constexpr int GC_W = GC_T / 64;       // waves: each takes every GC_W-th candidate
// BUILD = true (ransac_nb_build, once per call, right after ransac_init): the same tiles and
// windows, but instead of relabelling every found neighbour is appended to the point's list
// -- GC_W sub-lists of NB_SUB entries per point, sub-list g written by the wave that tests
// every GC_W-th candidate from g on; position deltas, 2 bytes each. The sweeps of every
// round and the joint refinement walk those lists (100-300 gathers per point on EPOS's
// many-to-many sets) instead of repeating ~1 100 pair tests per point and sweep; results are
// the same integers. A slot in which some sub-list does not fit (more than NB_SUB entries, a
// delta beyond 16 bits) keeps nb_ok = 0 and its sweeps scan the windows as before.
template <bool BUILD>
__global__ __launch_bounds__(GC_T) void ransac_gc_sweep(
    const double* __restrict__ xy_all, const double* __restrict__ xyz_all,
    const int64_t* __restrict__ slot_base, EposFitParams prm, Work w,
    const uint8_t* __restrict__ lab_in_all, uint8_t* __restrict__ lab_out_all) {
  const int s = blockIdx.y;
  if (BUILD ? w.done[s] != 0 : w.state[s] != 1) return;
  // one 48-byte record per candidate = three ds_read_b128 (round 3; eight separate
  // arrays before: the reads of a candidate were serialised behind one another, ~1000
  // cycles per candidate and wave -- profiles/r03/gc_sweep_ablation.txt)
  __shared__ __attribute__((aligned(16))) GcCand c_rec[GC_T];
  __shared__ int64_t s_win[2];
  __shared__ int s_deg[GC_W][64], s_n0[GC_W][64];
  __shared__ int64_t s_S[GC_W][64];
  __shared__ int s_cnt4[GC_W];
  const int t = threadIdx.x, pt = t & 63, sub = t >> 6;
  const int64_t base = slot_base[s];
  const int64_t n = slot_base[s + 1] - base;
  const double* xy = xy_all + 2 * base;
  const double* xyz = xyz_all + 3 * base;
  const uint8_t* lab_in = BUILD ? nullptr : lab_in_all + base;
  uint8_t* lab_out = BUILD ? nullptr : lab_out_all + base;
  const int32_t* gq = w.gq + base;
  const int32_t* yorder = w.yorder ? w.yorder + base : nullptr;
  uint16_t* nb_cnt = w.nb_cnt + base * GC_W;
  int16_t* nb_pool = w.nb_pool + base * (GC_W * NB_SUB);
  const bool use_lists = !BUILD && w.nb_ok[s] != 0;        // uniform over the slot
  const double lam = prm.spatial_coherence_weight, rad = prm.neighborhood_ball_radius;
  const double s2 = prm.scaling_from_millimeters * prm.scaling_from_millimeters;
  const double r2 = rad * rad;
  for (int64_t tile = blockIdx.x; tile * 64 < n; tile += gridDim.x) {
    const int64_t pos0 = tile * 64;
    const int64_t pos = pos0 + pt;
    const bool valid = pos < n;
    const int32_t p = valid ? (yorder ? yorder[pos] : static_cast<int32_t>(pos)) : 0;
    const uint8_t lp = BUILD ? 0 : (valid ? lab_in[p] : 2);
    const bool act = BUILD ? valid : lp != 2;
    int deg = 0, n0 = 0;
    int64_t S = 0;
    if (use_lists) {
      // ---- the point's neighbour list: this wave walks the sub-list it wrote ----------
      if (valid && act) {
        const int cntp = nb_cnt[pos * GC_W + sub];
        const int16_t* lst = nb_pool + (pos * GC_W + sub) * NB_SUB;
        for (int i0 = 0; i0 < cntp; i0 += 4) {
          int32_t o[4];
          int ok4[4], lo4[4], q4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            ok4[u] = i < cntp;
            const int64_t op = pos + lst[ok4[u] ? i : 0];
            o[u] = yorder ? yorder[op] : static_cast<int32_t>(op);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) { lo4[u] = lab_in[o[u]]; q4[u] = gq[o[u]]; }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int nb = ok4[u] & static_cast<int>(lo4[u] != 2);
            deg += nb;
            S += nb ? q4[u] : 0;
            n0 += nb & static_cast<int>(lo4[u] == 0);
          }
        }
      }
    } else {
    const double px = xy[2 * p], py = xy[2 * p + 1];
    const double pX = xyz[3 * p], pY = xyz[3 * p + 1], pZ = xyz[3 * p + 2];
    int nlist = 0;                 // BUILD: entries in this thread's sub-list
    bool ovf = false;
    {
      // window of sorted positions whose row can hold a neighbour of a point of the tile:
      // GC_T-ary searches (every thread probes one position per step: two or three
      // dependent loads instead of the ~13 of a binary search by one thread)
      const int64_t last = pos0 + 63 < n ? pos0 + 63 : n - 1;
      const int32_t pf = yorder ? yorder[pos0] : static_cast<int32_t>(pos0);
      const int32_t pl = yorder ? yorder[last] : static_cast<int32_t>(last);
      const double ylo = xy[2 * pf + 1] - rad, yhi = xy[2 * pl + 1] + rad;
      for (int side = 0; side < 2; ++side) {
        // side 0: first position in [0, pos0] with y >= ylo; side 1: first position in
        // [last + 1, n] with y > yhi   (y is non-decreasing along the sorted order)
        int64_t lo = side ? last + 1 : 0, hi = side ? n : pos0;
        while (hi - lo > 0) {
          const int64_t step = (hi - lo + GC_T - 1) / GC_T;
          const int64_t q = lo + static_cast<int64_t>(t) * step;
          bool pred = false;                       // "position q is at or beyond the bound"
          if (q < hi) {
            const int32_t o = yorder ? yorder[q] : static_cast<int32_t>(q);
            const double yq = xy[2 * o + 1];
            pred = side ? yq > yhi : yq >= ylo;
          }
          // first thread whose probe satisfies the predicate (monotone along t)
          const unsigned long long b = __ballot(pred);
          if ((t & 63) == 0) s_cnt4[t >> 6] = b ? (t + __ffsll(static_cast<long long>(b)) - 1) : GC_T;
          __syncthreads();
          int first = GC_T;
#pragma unroll
          for (int g = 0; g < GC_W; ++g) first = s_cnt4[g] < first ? s_cnt4[g] : first;
          __syncthreads();
          if (first == GC_T) {       // no probe at or beyond the bound: it lies after the
            const int64_t tv = (hi - lo - 1) / step;            // last probe made
            lo = lo + tv * step + 1;
          } else {                  // the bound lies in (probe[first - 1], probe[first]]
            const int64_t qf = lo + static_cast<int64_t>(first) * step;
            lo = first == 0 ? lo : lo + static_cast<int64_t>(first - 1) * step + 1;
            hi = qf;
          }
        }
        if (t == 0) s_win[side] = lo;
      }
    }
    __syncthreads();
    const int64_t wlo = s_win[0], whi = s_win[1];
#ifdef EPOS_GC_STATS
    if (t == 0) {
      atomicAdd(&g_gc_stats[0], 1ull);
      atomicAdd(&g_gc_stats[1], static_cast<unsigned long long>(whi - wlo));
      atomicAdd(&g_gc_stats[2], static_cast<unsigned long long>(n));
    }
#endif
    for (int64_t c0 = wlo; c0 < whi; c0 += GC_T) {
      const int64_t c = c0 + t;
      if (c < whi) {
        const int32_t o = yorder ? yorder[c] : static_cast<int32_t>(c);
        GcCand r;
        r.x = xy[2 * o]; r.y = xy[2 * o + 1];
        r.X = xyz[3 * o]; r.Y = xyz[3 * o + 1]; r.Z = xyz[3 * o + 2];
        r.q = BUILD ? 0 : gq[o];
        r.ol = o | (static_cast<int32_t>(BUILD ? 0 : lab_in[o]) << 30);
        c_rec[t] = r;
      }
      __syncthreads();
      const int cnt = whi - c0 < GC_T ? static_cast<int>(whi - c0) : GC_T;
      // branch-free, four candidates per trip: their twelve 16-byte LDS reads (wave-uniform
      // addresses: broadcasts) are issued together from clamped positions, the tests are
      // arithmetic. (The first version -- early-outs between dependent LDS reads -- ran at
      // ~1500 cycles per candidate; eight scalar arrays still at ~1000.)
#ifdef EPOS_GC_ABL_NOLOOP       // ablation (tools/): everything but the pair tests
      if (cnt < 0)
#endif
      for (int j0 = sub; j0 < cnt; j0 += 4 * GC_W) {
        GcCand r[4];
        int okj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = j0 + u * GC_W;
          okj[u] = j < cnt;
          r[u] = c_rec[okj[u] ? j : sub];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int lo_ = (r[u].ol >> 30) & 3;
          const int32_t co = r[u].ol & 0x3fffffff, cq = r[u].q;
          const double dx = px - r[u].x, dy = py - r[u].y;
          const double dX = pX - r[u].X, dY = pY - r[u].Y, dZ = pZ - r[u].Z;
          const double d2 = (dx * dx + dy * dy) + s2 * ((dX * dX + dY * dY) + dZ * dZ);
          const int nb = okj[u] & static_cast<int>(act) & static_cast<int>(lo_ != 2) &
                         static_cast<int>(co != p) & static_cast<int>(d2 <= r2);
          if (BUILD) {
            if (nb) {              // position of the candidate: c0 + j
              const int64_t dl = c0 + (j0 + u * GC_W) - pos;
              if (nlist < NB_SUB && dl >= -32768 && dl <= 32767)
                nb_pool[(pos * GC_W + sub) * NB_SUB + nlist] = static_cast<int16_t>(dl);
              else
                ovf = true;
              ++nlist;
            }
          } else {
            deg += nb;
            S += nb ? cq : 0;
            n0 += nb & static_cast<int>(lo_ == 0);
          }
        }
      }
      __syncthreads();
    }
    if (BUILD) {
      if (valid) nb_cnt[pos * GC_W + sub] = static_cast<uint16_t>(ovf ? 0 : nlist);
      if (__any(valid && ovf) && pt == 0)
        __hip_atomic_store(w.nb_ok + s, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      continue;
    }
    }   // !use_lists
    s_deg[sub][pt] = deg; s_n0[sub][pt] = n0; s_S[sub][pt] = S;
    __syncthreads();
    if (sub == 0 && valid) {
      if (!act) {
        lab_out[p] = 2;
      } else {
        int64_t dg = 0, z0 = 0, Ss = 0;
#pragma unroll
        for (int g = 0; g < GC_W; ++g) { dg += s_deg[g][pt]; z0 += s_n0[g][pt]; Ss += s_S[g][pt]; }
        const int64_t qp = gq[p];
        const int64_t T = 2 * static_cast<int64_t>(GC_Q) * z0 - (dg * qp + Ss);
        const int64_t u = qp < GC_Q ? -2 * (static_cast<int64_t>(GC_Q) - qp)
                                    : 2 * static_cast<int64_t>(GC_Q);
        const double val = (1.0 - lam) * static_cast<double>(u) + lam * static_cast<double>(T);
        lab_out[p] = val < 0.0 ? 1 : 0;
      }
    }
    __syncthreads();
  }
}

// ---- the same sweep with the candidates in SCALAR registers (round 3). A candidate is the
// same for all points of a tile, so it does not belong in vector registers or LDS at all:
// every wave walks a contiguous share (1 / GS_W) of the window through s_load (32 bytes of
// static geometry from w.geo + 16 bytes of GcDyn per candidate, position ordered) and the
// fifteen fp64 operations of the distance take the candidate's fields as SGPR operands. No
// staging through LDS, no barriers inside the window, no per-candidate index arithmetic; "not
// active" is Z = +inf instead of a label test, the point itself is counted like any other
// candidate and taken out once at the end, the residual sum runs in 32 bits per block of
// GS_BLOCK candidates. The scalar cache streams poorly (every line is a miss to L2 and few
// misses are in flight: GS_P = 1 ran at ~0.5 bytes per ns and CU pair), so a lane holds
// GS_P points -- a workgroup covers GS_P adjacent tiles, whose windows are nearly the same
// rows -- and every loaded candidate is tested GS_P times. The waves' partial counts meet in
// LDS by integer atomics (exact, order free). Same integers as ransac_gc_sweep<false>
// (tests/test_gpu_fit_lists.py runs both).
constexpr int GS_W = EPOS_GS_WAVES;      // waves per workgroup: shares of the window
constexpr int GS_P = EPOS_GS_POINTS;     // points per lane: adjacent tiles per workgroup
constexpr int GS_T = GS_W * 64;
constexpr int GS_BLOCK = 2048;           // 2048 x 2^20 < 2^32: the 32-bit sum cannot wrap

__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32));
  return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}

struct __attribute__((aligned(32))) GcGeo { double x, y, X, Y; };
typedef double GsGeoV __attribute__((ext_vector_type(4)));      // x, y, X, Y
typedef int32_t GsDynV __attribute__((ext_vector_type(4)));     // Z (two words), q, out

__global__ __launch_bounds__(GS_T) void ransac_gc_scan(
    const int64_t* __restrict__ slot_base, EposFitParams prm, Work w,
    const GcGeo* __restrict__ geo_all, const GcDyn* __restrict__ dyn_in_all,
    GcDyn* __restrict__ dyn_out_all, const uint8_t* __restrict__ lab_in_all,
    uint8_t* __restrict__ lab_out_all, int8_t* __restrict__ flip_out_all) {
  const int s = blockIdx.y;
  if (w.state[s] != 1) return;
  __shared__ unsigned s_deg[GS_P * 64], s_n0[GS_P * 64];
  __shared__ unsigned long long s_S[GS_P * 64];
  const int t = threadIdx.x, pt = t & 63;
  const int sub = __builtin_amdgcn_readfirstlane(t >> 6);
  const int64_t base = slot_base[s];
  const int64_t n = slot_base[s + 1] - base;
  const GcGeo* __restrict__ geo = geo_all + base;
  const GcDyn* __restrict__ dyn_in = dyn_in_all + base;
  GcDyn* __restrict__ dyn_out = dyn_out_all + base;
  const uint8_t* lab_in = lab_in_all + base;
  uint8_t* lab_out = lab_out_all + base;
  int8_t* __restrict__ flip_out = flip_out_all + base;
  const int32_t* yorder = w.yorder ? w.yorder + base : nullptr;
  const double lam = prm.spatial_coherence_weight, rad = prm.neighborhood_ball_radius;
  const double s2 = prm.scaling_from_millimeters * prm.scaling_from_millimeters;
  const double r2 = rad * rad;
  const GsGeoV* __restrict__ gv = reinterpret_cast<const GsGeoV*>(geo);
  const GsDynV* __restrict__ dv = reinterpret_cast<const GsDynV*>(dyn_in);
  const int64_t clast = n - 1;
  const int64_t tiles = (n + 63) / 64;
  FIT_TRACE(3, s == 0 && blockIdx.x == 0 && t == 0);
  for (int64_t grp = blockIdx.x; grp * GS_P < tiles; grp += gridDim.x) {
    const int64_t tile0 = grp * GS_P;
    const int64_t tile1 = tile0 + GS_P - 1 < tiles - 1 ? tile0 + GS_P - 1 : tiles - 1;
    const int64_t pos0 = tile0 * 64;
    // the group's candidate window: first tile's lower bound .. last tile's upper bound
    // (found once per call by ransac_init; the bounds are monotone along the order)
    const int32_t* wn = w.win + 2 * (base / 64 + s);
    const int64_t wlo = uniform64(wn[2 * tile0]), whi = uniform64(wn[2 * tile1 + 1]);
    for (int i = t; i < GS_P * 64; i += GS_T) { s_deg[i] = 0u; s_n0[i] = 0u; s_S[i] = 0ull; }
#ifdef EPOS_GC_STATS
    if (t == 0) {
      atomicAdd(&g_gc_stats[0], 1ull);
      atomicAdd(&g_gc_stats[1], static_cast<unsigned long long>(whi - wlo));
      atomicAdd(&g_gc_stats[2], static_cast<unsigned long long>(n));
    }
#endif
    double px[GS_P], py[GS_P], pX[GS_P], pY[GS_P], pZ[GS_P];
    uint32_t deg[GS_P], n0[GS_P], S32[GS_P];
    unsigned long long S[GS_P];
#pragma unroll
    for (int j = 0; j < GS_P; ++j) {
      const int64_t pos = pos0 + j * 64 + pt;
      const int64_t posc = pos < n ? pos : clast;
      const GsGeoV mg = gv[posc];
      const GsDynV md = dv[posc];
      px[j] = mg.x; py[j] = mg.y; pX[j] = mg.z; pY[j] = mg.w;
      pZ[j] = __builtin_bit_cast(double, md.xy);
      deg[j] = 0; n0[j] = 0; S32[j] = 0; S[j] = 0;
    }
    __syncthreads();                                 // the LDS counters are zero
    FIT_TRACE(3, s == 0 && blockIdx.x == 0 && t == 0);
    const int64_t chunk = (whi - wlo + GS_W - 1) / GS_W;
    const int64_t ca = wlo + sub * chunk;
    const int64_t cb = ca + chunk < whi ? ca + chunk : whi;
// one candidate: its geometry (x, y, X, Y) and dynamic record (Z | q, out) arrive as whole
// 32- and 16-byte scalar loads and are tested against the lane's GS_P points
#define EPOS_GS_TEST(G, D)                                                                \
    {                                                                                     \
      const double cZ = __builtin_bit_cast(double, (D).xy);                               \
      _Pragma("unroll") for (int j = 0; j < GS_P; ++j) {                                  \
        const double dx = px[j] - (G).x, dy = py[j] - (G).y;                              \
        const double dX = pX[j] - (G).z, dY = pY[j] - (G).w, dZ = pZ[j] - cZ;             \
        const double d2 = (dx * dx + dy * dy) + s2 * ((dX * dX + dY * dY) + dZ * dZ);     \
        const uint32_t nb = d2 <= r2 ? 1u : 0u;                                           \
        deg[j] += nb;                                                                     \
        S32[j] += __umul24(nb, static_cast<uint32_t>((D).z));                             \
        n0[j] += __umul24(nb, static_cast<uint32_t>((D).w));                              \
      }                                                                                   \
    }
    for (int64_t c0 = ca; c0 < cb; c0 += GS_BLOCK) {
      const int64_t c1 = c0 + GS_BLOCK < cb ? c0 + GS_BLOCK : cb;
      int64_t c = c0;
      // two candidates in flight while two are tested (reads past the block are clamped to
      // the slot and never tested)
      GsGeoV ga0 = gv[c < clast ? c : clast], ga1 = gv[c + 1 < clast ? c + 1 : clast];
      GsDynV da0 = dv[c < clast ? c : clast], da1 = dv[c + 1 < clast ? c + 1 : clast];
#if defined(EPOS_GS_ABL_NOLOAD)   // ablation (tools/): the arithmetic alone, one candidate reused
      const GsGeoV gb0 = gv[c + 2 < clast ? c + 2 : clast], gb1 = gv[c + 3 < clast ? c + 3 : clast];
      const GsDynV db0 = dv[c + 2 < clast ? c + 2 : clast], db1 = dv[c + 3 < clast ? c + 3 : clast];
      for (; c + 4 <= c1; c += 4) {
        EPOS_GS_TEST(ga0, da0)
        EPOS_GS_TEST(ga1, da1)
        EPOS_GS_TEST(gb0, db0)
        EPOS_GS_TEST(gb1, db1)
        asm volatile("" ::: "memory");
      }
#elif defined(EPOS_GS_ABL_NOMATH)  // ablation: the scalar loads alone
      for (; c + 4 <= c1; c += 4) {
        const GsGeoV gb0 = gv[c + 2], gb1 = gv[c + 3];
        const GsDynV db0 = dv[c + 2], db1 = dv[c + 3];
        deg[0] += (ga0.x == 1e300 ? 1u : 0u) + (ga1.x == 1e300 ? 1u : 0u) + (da0.z == -5 ? 1u : 0u) + (da1.z == -5 ? 1u : 0u);
        const int64_t e0 = c + 4 < clast ? c + 4 : clast, e1 = c + 5 < clast ? c + 5 : clast;
        ga0 = gv[e0]; ga1 = gv[e1]; da0 = dv[e0]; da1 = dv[e1];
        deg[0] += (gb0.x == 1e300 ? 1u : 0u) + (gb1.x == 1e300 ? 1u : 0u) + (db0.z == -5 ? 1u : 0u) + (db1.z == -5 ? 1u : 0u);
      }
#else
      for (; c + 4 <= c1; c += 4) {
        const GsGeoV gb0 = gv[c + 2], gb1 = gv[c + 3];
        const GsDynV db0 = dv[c + 2], db1 = dv[c + 3];
        EPOS_GS_TEST(ga0, da0)
        EPOS_GS_TEST(ga1, da1)
        const int64_t e0 = c + 4 < clast ? c + 4 : clast, e1 = c + 5 < clast ? c + 5 : clast;
        ga0 = gv[e0]; ga1 = gv[e1]; da0 = dv[e0]; da1 = dv[e1];
        EPOS_GS_TEST(gb0, db0)
        EPOS_GS_TEST(gb1, db1)
      }
#endif
      if (c < c1) {                       // up to three left: ga0 / ga1 hold c, c + 1
        EPOS_GS_TEST(ga0, da0)
        if (c + 1 < c1) EPOS_GS_TEST(ga1, da1)
        if (c + 2 < c1) { const GsGeoV g2 = gv[c + 2]; const GsDynV d2_ = dv[c + 2]; EPOS_GS_TEST(g2, d2_) }
      }
#pragma unroll
      for (int j = 0; j < GS_P; ++j) { S[j] += S32[j]; S32[j] = 0; }
    }
#undef EPOS_GS_TEST
    FIT_TRACE(3, s == 0 && blockIdx.x == 0 && t == 0);
#pragma unroll
    for (int j = 0; j < GS_P; ++j) {
      atomicAdd(&s_deg[j * 64 + pt], deg[j]);
      atomicAdd(&s_n0[j * 64 + pt], n0[j]);
      atomicAdd(&s_S[j * 64 + pt], S[j]);
    }
    __syncthreads();
    // wave j (when there are fewer waves than tiles: j, j + GS_W, ...) decides tile j
    for (int j = sub; j < GS_P; j += GS_W) {
      const int64_t pos = pos0 + j * 64 + pt;
      if (pos >= n) continue;
      const GsGeoV mg = gv[pos];
      const GsDynV md = dv[pos];
      const int32_t p = yorder ? yorder[pos] : static_cast<int32_t>(pos);
      const double mZ = __builtin_bit_cast(double, md.xy);
      GcDyn nd;
      nd.Z = mZ; nd.q = md.z; nd.out = 0;
      int8_t flip = 0;
      GcAcc raw;
      raw.deg = s_deg[j * 64 + pt]; raw.n0 = s_n0[j * 64 + pt]; raw.S = s_S[j * 64 + pt];
      w.acc[base + pos] = raw;
      if (lab_in[p] == 2) {
        lab_out[p] = 2;
      } else {
        // the point met itself in the window: take it out again (the same test on itself)
        const double dx = mg.x - mg.x, dy = mg.y - mg.y;
        const double dX = mg.z - mg.z, dY = mg.w - mg.w, dZ = mZ - mZ;
        const double d2 = (dx * dx + dy * dy) + s2 * ((dX * dX + dY * dY) + dZ * dZ);
        const int64_t self = d2 <= r2 ? 1 : 0;
        const int64_t qp = md.z;
        const int64_t dg = static_cast<int64_t>(s_deg[j * 64 + pt]) - self;
        const int64_t z0 = static_cast<int64_t>(s_n0[j * 64 + pt]) - self * md.w;
        const int64_t Ss = static_cast<int64_t>(s_S[j * 64 + pt]) - self * qp;
        const int64_t T = 2 * static_cast<int64_t>(GC_Q) * z0 - (dg * qp + Ss);
        const int64_t u = qp < GC_Q ? -2 * (static_cast<int64_t>(GC_Q) - qp)
                                    : 2 * static_cast<int64_t>(GC_Q);
        const double val = (1.0 - lam) * static_cast<double>(u) + lam * static_cast<double>(T);
        const int inl = val < 0.0 ? 1 : 0;
        lab_out[p] = static_cast<uint8_t>(inl);
        nd.out = 1 - inl;
        flip = static_cast<int8_t>(nd.out - md.w);
      }
      dyn_out[pos] = nd;
      flip_out[pos] = flip;
    }
    __syncthreads();
    FIT_TRACE(3, s == 0 && blockIdx.x == 0 && t == 0);
  }
}

// ---- every FURTHER sweep of a round: the labels changed only at the positions the previous
// sweep flipped, and of a point's three sums only the count of outlier-labelled neighbours
// depends on labels at all. So the sweep after a full scan tests every point against the
// FLIPPED candidates of its window alone (typically a few percent of it):
//   raw_n0' = raw_n0 + sum over flipped neighbours of (+1: became outlier, -1: became inlier)
// -- integers, hence the same labels as another full scan (tests/test_gpu_fit_lists.py). One
// workgroup per tile: the window's flip bytes are read in chunks of 256 positions, the
// flipped candidates of a chunk are compacted into LDS with their geometry, wave g tests
// entries g, g + 4, ...
struct GdCand { double x, y, X, Y, Z; int32_t sign, pad; };

__global__ __launch_bounds__(256) void ransac_gc_delta(
    const int64_t* __restrict__ slot_base, EposFitParams prm, Work w,
    const GcGeo* __restrict__ geo_all, const GcDyn* __restrict__ dyn_all,
    const int8_t* __restrict__ flip_in_all, int8_t* __restrict__ flip_out_all,
    const uint8_t* __restrict__ lab_in_all, uint8_t* __restrict__ lab_out_all) {
  const int s = blockIdx.y;
  if (w.state[s] != 1) return;
  __shared__ __attribute__((aligned(16))) GdCand s_list[256];
  __shared__ int s_wcnt[4];
  __shared__ int s_dn0[64];
  const int t = threadIdx.x, pt = t & 63, sub = t >> 6;
  const int64_t base = slot_base[s];
  const int64_t n = slot_base[s + 1] - base;
  const GcGeo* __restrict__ geo = geo_all + base;
  const GcDyn* __restrict__ dyn = dyn_all + base;       // Z and q: the round's dyn_a
  const int8_t* __restrict__ flip_in = flip_in_all + base;
  int8_t* __restrict__ flip_out = flip_out_all + base;
  const uint8_t* lab_in = lab_in_all + base;
  uint8_t* lab_out = lab_out_all + base;
  GcAcc* acc = w.acc + base;
  const int32_t* yorder = w.yorder ? w.yorder + base : nullptr;
  const double lam = prm.spatial_coherence_weight, rad = prm.neighborhood_ball_radius;
  const double s2 = prm.scaling_from_millimeters * prm.scaling_from_millimeters;
  const double r2 = rad * rad;
  FIT_TRACE(5, s == 0 && blockIdx.x == 0 && t == 0);
  for (int64_t tile = blockIdx.x; tile * 64 < n; tile += gridDim.x) {
    const int64_t pos = tile * 64 + pt;
    const bool valid = pos < n;
    const int64_t posc = valid ? pos : n - 1;
    const GcGeo me = geo[posc];
    const GcDyn md = dyn[posc];
    const int32_t* wn = w.win + 2 * (base / 64 + s + tile);
    const int64_t wlo = wn[0], whi = wn[1];
    if (t < 64) s_dn0[t] = 0;
    int dn0 = 0;
    for (int64_t c0 = wlo; c0 < whi; c0 += 256) {
      const int64_t c = c0 + t;
      const int f = c < whi ? flip_in[c] : 0;
      const unsigned long long b = __ballot(f != 0);
      if (pt == 0) s_wcnt[sub] = __popcll(b);
      __syncthreads();                               // also: the list of the last chunk is done
      int off = 0, cnt = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) { off += g < sub ? s_wcnt[g] : 0; cnt += s_wcnt[g]; }
      if (f != 0) {
        const GcGeo cg = geo[c];
        GdCand e;
        e.x = cg.x; e.y = cg.y; e.X = cg.X; e.Y = cg.Y; e.Z = dyn[c].Z; e.sign = f; e.pad = 0;
        s_list[off + __popcll(b & ((1ull << pt) - 1ull))] = e;
      }
      __syncthreads();
      for (int e = sub; e < cnt; e += 4) {
        const GdCand r = s_list[e];
        const double dx = me.x - r.x, dy = me.y - r.y;
        const double dX = me.X - r.X, dY = me.Y - r.Y, dZ = md.Z - r.Z;
        const double d2 = (dx * dx + dy * dy) + s2 * ((dX * dX + dY * dY) + dZ * dZ);
        dn0 += d2 <= r2 ? r.sign : 0;
      }
      __syncthreads();
    }
    if (dn0 != 0) atomicAdd(&s_dn0[pt], dn0);
    __syncthreads();
    if (sub == 0 && valid) {
      const int32_t p = yorder ? yorder[pos] : static_cast<int32_t>(pos);
      const int prev = lab_in[p];
      int8_t flip = 0;
      if (prev == 2) {
        lab_out[p] = 2;
      } else {
        GcAcc raw = acc[pos];
        raw.n0 = static_cast<uint32_t>(static_cast<int>(raw.n0) + s_dn0[pt]);
        acc[pos] = raw;                                // the next sweep continues from here
        const int out_prev = 1 - prev;                 // label 1 = inlier
        const double dx = me.x - me.x, dy = me.y - me.y;
        const double dX = me.X - me.X, dY = me.Y - me.Y, dZ = md.Z - md.Z;
        const double d2 = (dx * dx + dy * dy) + s2 * ((dX * dX + dY * dY) + dZ * dZ);
        const int64_t self = d2 <= r2 ? 1 : 0;
        const int64_t qp = md.q;
        const int64_t dg = static_cast<int64_t>(raw.deg) - self;
        const int64_t z0 = static_cast<int64_t>(raw.n0) - self * out_prev;
        const int64_t Ss = static_cast<int64_t>(raw.S) - self * qp;
        const int64_t T = 2 * static_cast<int64_t>(GC_Q) * z0 - (dg * qp + Ss);
        const int64_t u = qp < GC_Q ? -2 * (static_cast<int64_t>(GC_Q) - qp)
                                    : 2 * static_cast<int64_t>(GC_Q);
        const double val = (1.0 - lam) * static_cast<double>(u) + lam * static_cast<double>(T);
        const int inl = val < 0.0 ? 1 : 0;
        lab_out[p] = static_cast<uint8_t>(inl);
        flip = static_cast<int8_t>((1 - inl) - out_prev);
      }
      flip_out[pos] = flip;
    }
    __syncthreads();
  }
  FIT_TRACE(5, s == 0 && blockIdx.x == 0 && t == 0);
}

// ---- round, step 4: local optimisation (ii) on the labelled inliers, the instance
// acceptance tests, labelling + removal of the explained correspondences.
__global__ __launch_bounds__(256) void ransac_refit_accept(
    const double* __restrict__ xy_all, const double* __restrict__ xyz_all,
    const int64_t* __restrict__ slot_base, const double* __restrict__ Ks,
    const int32_t* __restrict__ max_models, EposFitParams prm, int max_k, Work w,
    const uint8_t* __restrict__ lab_all, double* poses, double* scores,
    int32_t* num_models, int32_t* labels_all, int lo_launch, int n_lo) {
  const int s = blockIdx.y;
  const int g = blockIdx.x;          // LO_G workgroups refit together; workgroup 0 accepts
  const int t = threadIdx.x;
  const int state = w.state[s];
  FIT_TRACE(2, s == 0 && g == 0 && t == 0);
  if (state == 0) return;                                  // uniform over the slot
  int want = max_models[s];
  if (want < 0 || want > max_k) want = max_k;
  const int k = num_models[s];
  const int64_t base = slot_base[s];
  const int64_t n = slot_base[s + 1] - base;
  int32_t* active = w.active + base;
  const int64_t n_active = w.n_active[s];
  __shared__ LoLds s_lo;
  __shared__ int s_inl[4], s_new[4];
  const int lane = t & 63, wave = t >> 6;
  const double* xy = xy_all + 2 * base;
  const double* xyz = xyz_all + 3 * base;
  int32_t* labels = labels_all + base;
  double K[9];
  for (int i = 0; i < 9; ++i) K[i] = Ks[s * 9 + i];
  const double thr2 = prm.threshold * prm.threshold;
  double pose[12];
  for (int i = 0; i < 12; ++i) pose[i] = w.cur_pose[s * 12 + i];
  double best_score = w.cur_score[s];
  int best_count = w.cur_count[s];
  if (state == 1) {
    const uint8_t* lab = lab_all + base;
    LoSync sy;
    sy.cnt = w.lo_cnt + static_cast<int64_t>(s) * n_lo + lo_launch;
    sy.data = w.lo_data + static_cast<int64_t>(s) * (2 * LO_G * 4 * LO_NV);
    sy.timeout = w.lo_timeout;
    sy.spin_max = w.lo_spin_max;
    sy.g = g;
    sy.epoch = 0;
    // the first step on the labelled set starts from the accepted pose (its score is
    // known); from then on each pass scores a candidate and steps from it
    double cand[12];
    const int32_t* idx = n_active == n ? nullptr : active;
    FIT_TRACE(2, s == 0 && g == 0 && t == 0);
    int fail = lo_pass<false>(pose, K, xy, xyz, idx, n_active, thr2, t, sy, &s_lo, lab, 1,
                              nullptr, nullptr, cand);
    FIT_TRACE(2, s == 0 && g == 0 && t == 0);
    for (int li = 0; li < prm.lo_iters && !fail; ++li) {
      double sc, cand2[12];
      int cnt;
      const int fail2 = lo_pass<true>(cand, K, xy, xyz, idx, n_active, thr2, t, sy, &s_lo,
                                      lab, 1, &sc, &cnt, cand2);
      FIT_TRACE(2, s == 0 && g == 0 && t == 0);
      if (!(sc > best_score)) break;
      const double gain = sc - best_score;
      best_score = sc; best_count = cnt;
      for (int i = 0; i < 12; ++i) { pose[i] = cand[i]; cand[i] = cand2[i]; }
      fail = fail2;
      if (!(gain > LO_MIN_GAIN * sc)) break;
    }
  }
  // A hand-off that timed out (a sibling workgroup was not scheduled for ~0.5 s: never seen,
  // but then the sums -- and everything derived from them -- are garbage) is reported, not
  // hidden: the slot's instance count becomes -1 and the host entry points raise.
  if (__hip_atomic_load(w.lo_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    if (g == 0 && t == 0) { num_models[s] = -1; w.done[s] = 1; }
    return;
  }
  if (g != 0) return;          // the siblings only helped with the sums
  FIT_TRACE(2, s == 0 && t == 0);
  if (best_count < prm.min_point_number) {
    if (t == 0) round_failed(s, w, prm, want, k, n_active);
    return;
  }
  // ---- inliers over ALL correspondences of the slot -> bitset (chunk c: wave c % 4);
  //      four chunks per trip: their loads are issued together (one dependent round trip
  //      per trip instead of per chunk -- this loop is latency, not arithmetic)
  const int64_t words = (n + 63) / 64;
  const int64_t wbase = base / 64 + s;
  uint64_t* cur = w.inl_bits + static_cast<int64_t>(k) * w.words_total + wbase;
  int n_inl = 0, n_new = 0;
  for (int64_t c0 = wave; c0 < words; c0 += 16) {
    double x2[4][2], x3[4][3];
    int32_t lb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = (c0 + 4 * u) * 64 + lane;
      const int64_t ic = i < n ? i : n - 1;
      x2[u][0] = xy[2 * ic]; x2[u][1] = xy[2 * ic + 1];
      x3[u][0] = xyz[3 * ic]; x3[u][1] = xyz[3 * ic + 1]; x3[u][2] = xyz[3 * ic + 2];
      lb[u] = labels[ic];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t c = c0 + 4 * u;
      if (c >= words) break;                               // wave-uniform
      const int64_t i = c * 64 + lane;
      bool inl = false;
      double e2, Xc[3], r[2];
      if (i < n && !reproj(pose, K, x2[u], x3[u], &e2, Xc, r)) inl = e2 < thr2;
      const bool fresh = inl && lb[u] < 0;
      const uint64_t bits = __ballot(inl);
      n_inl += __popcll(bits);
      n_new += __popcll(__ballot(fresh));
      if (lane == 0) cur[c] = bits;
    }
  }
  __shared__ int s_ok;
  __shared__ int s_keep[4][4];
  if (lane == 0) { s_inl[wave] = n_inl; s_new[wave] = n_new; }
  __threadfence_block();
  __syncthreads();
  FIT_TRACE(2, s == 0 && t == 0);
  n_inl = s_inl[0] + s_inl[1] + s_inl[2] + s_inl[3];
  n_new = s_new[0] + s_new[1] + s_new[2] + s_new[3];
  if (wave == 0) {                                         // the acceptance tests
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
    if (lane == 0) s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  FIT_TRACE(2, s == 0 && t == 0);
  if (!s_ok) { if (t == 0) round_failed(s, w, prm, want, k, n_active); return; }
  // ---- accept: write the pose, label + remove its inliers. Stable compaction of the
  //      active list IN PLACE by the whole workgroup, 1024 entries per trip: every entry
  //      of the trip is read before any is written, and a trip only writes below its own
  //      start + what it kept (positions that have all been read).
  if (t < 12) poses[(static_cast<int64_t>(s) * max_k + k) * 12 + t] = pose[t];
  if (t == 0) scores[static_cast<int64_t>(s) * max_k + k] = best_score;
  if (k + 1 >= want) {
    // the slot's LAST instance: nobody reads its active list again, so the compaction (a
    // dependent gather per entry) is skipped -- the unexplained inliers are labelled
    // straight from the bitset, coalesced. (Every inlier that is still unlabelled IS
    // active: the two sets are kept identical by the compaction below.)
    for (int64_t i = t; i < n; i += 256)
      if (((cur[i >> 6] >> (i & 63)) & 1ull) && labels[i] < 0) labels[i] = k;
    if (t == 0) {
      w.tries[s] = 0;
      w.last_new[s] = n_new;
      num_models[s] = k + 1;
      w.done[s] = 1;
    }
    FIT_TRACE(2, s == 0 && t == 0);
    return;
  }
  int64_t wpos = 0;
  for (int64_t c0 = 0; c0 < n_active; c0 += 1024) {
    int32_t pv[4];
    bool keep[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = c0 + u * 256 + t;
      pv[u] = i < n_active ? active[i] : 0;
    }
    uint64_t cw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) cw[u] = cur[pv[u] >> 6];
    int rank[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = c0 + u * 256 + t;
      const bool inl = i < n_active && ((cw[u] >> (pv[u] & 63)) & 1ull);
      if (inl) labels[pv[u]] = k;
      keep[u] = i < n_active && !inl;
      const uint64_t kb = __ballot(keep[u]);
      rank[u] = __popcll(kb & ((1ull << lane) - 1ull));
      if (lane == 0) s_keep[u][wave] = __popcll(kb);
    }
    __syncthreads();                       // every entry of the trip has been read
    int total = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int g2 = 0; g2 < 4; ++g2) total += s_keep[u][g2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int off = 0;
#pragma unroll
      for (int uu = 0; uu < 4; ++uu)
#pragma unroll
        for (int g2 = 0; g2 < 4; ++g2)
          if (uu * 4 + g2 < u * 4 + wave) off += s_keep[uu][g2];
      if (keep[u]) active[wpos + off + rank[u]] = pv[u];
    }
    wpos += total;
    __syncthreads();                       // s_keep is reused by the next trip
  }
  if (t == 0) {
    w.n_active[s] = static_cast<int32_t>(wpos);
    w.tries[s] = 0;
    w.last_new[s] = n_new;
    num_models[s] = k + 1;
    if (k + 1 >= want) w.done[s] = 1;
  }
  FIT_TRACE(2, s == 0 && t == 0);
}

// ------------------------------------------------ joint refinement (PEARL's role) --
// For slots with 2 <= k <= max_model_number_for_optimization accepted instances, over
// ALL correspondences of the slot: labels {0..k-1, k = outlier}; data term
// D_p(m) = min(e_pm^2 / (1.5 tau)^2, 1), D_p(outlier) = (tau / 1.5 tau)^2; degree-
// normalised Potts smoothness (a point pays lambda times the fraction of its neighbours
// with another label); everything in 2^-20 fixed point (exact integer sums: order
// independent). One iteration = gc_sweeps synchronous
// relabelling sweeps + one Gauss-Newton refit of every instance on its points, kept iff
// the energy dropped (DESIGN.md "Pose fitting", step 6).
constexpr int PEARL_MAX_K = 8;
static_assert(PEARL_DT == PEARL_MAX_K + 1, "one data-term column per label incl. the outlier");
constexpr int PEARL_BINS = 64;    // bins per energy sum (the workgroups' atomics spread over them)
static_assert(4 * PEARL_BINS == 256, "pearl_begin zeroes the bins with one 256-thread workgroup");

__device__ __forceinline__ int64_t pearl_data_term(const double* pose, const double* K,
                                                   const double* xy2, const double* xyz3,
                                                   double tthr2) {
  double e2, Xc[3], r[2];
  if (reproj(pose, K, xy2, xyz3, &e2, Xc, r)) return GC_Q;
  double d = e2 / tthr2;
  if (!(d < 1.0)) d = 1.0;
  return static_cast<int64_t>(d * static_cast<double>(GC_Q));
}

__global__ void pearl_setup(const int32_t* num_models, EposFitParams prm, Work w, int S) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const int k = num_models[s];
  w.pearl_state[s] = (k >= 2 && k <= PEARL_MAX_K && k <= prm.max_model_number_for_optimization &&
                      prm.spatial_coherence_weight > 0.0 && prm.neighborhood_ball_radius > 0.0 &&
                      prm.gc_sweeps >= 1) ? 1 : 0;
}

__global__ __launch_bounds__(256) void pearl_begin(const int64_t* slot_base,
                                                   const int32_t* num_models, Work w,
                                                   const int32_t* labels_all) {
  const int s = blockIdx.y;
  if (!w.pearl_state[s]) return;
  const int64_t base = slot_base[s], n = slot_base[s + 1] - base;
  const int k = num_models[s];
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const int32_t l = labels_all[base + i];
    const int64_t pos = w.ypos ? w.ypos[base + i] : i;       // the sweeps' labels: position order
    w.lab_a[base + pos] = static_cast<uint8_t>(l >= 0 && l < k ? l : k);
  }
  if (blockIdx.x == 0) w.pearl_acc[s * (4 * PEARL_BINS) + threadIdx.x] = 0ull;   // 4 x 64 bins
  if (blockIdx.x == 0 && threadIdx.x == 0) w.pearl_moved[s] = 0;
}

// Data terms of every point of a slot against the k poses (which = 0: the accepted ones,
// 1: the candidates of pearl_refit) -> Work::pearl_dt, one thread per point. The sweeps and
// the energy pass then only walk neighbours: with the fp64 reprojection inside them, one
// point per wave, 61 of 64 lanes idled through it and its registers cost the walk three of
// eight waves per SIMD (79 us per sweep at C4's size, 40 for the walk alone, 12 for the
// terms alone; profiles/r04). Same pearl_data_term, same integers.
__global__ __launch_bounds__(256) void pearl_data(
    const double* __restrict__ xy_all, const double* __restrict__ xyz_all,
    const int64_t* __restrict__ slot_base, const double* __restrict__ Ks,
    const int32_t* __restrict__ num_models, EposFitParams prm, int max_k, Work w,
    const double* __restrict__ poses, int which) {
  const int s = blockIdx.y;
  if (!w.pearl_state[s]) return;
  if (which == 1 && !w.pearl_moved[s]) return;
  const int64_t base = slot_base[s], n = slot_base[s + 1] - base;
  const int k = num_models[s];
  const double* pp = which ? w.pearl_pose + static_cast<int64_t>(s) * PEARL_MAX_K * 12
                           : poses + static_cast<int64_t>(s) * max_k * 12;
  double K[9];
  for (int i = 0; i < 9; ++i) K[i] = Ks[s * 9 + i];
  const double tthr = 1.5 * prm.threshold, tthr2 = tthr * tthr;
  const double thr2 = prm.threshold * prm.threshold;
  const int32_t d_out = static_cast<int32_t>(
      static_cast<int64_t>((thr2 / tthr2) * static_cast<double>(GC_Q)));
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; p < n;
       p += static_cast<int64_t>(gridDim.x) * 256) {
    int32_t* row = w.pearl_dt + (base + p) * PEARL_DT;
    for (int m = 0; m < k; ++m)
      row[m] = static_cast<int32_t>(
          pearl_data_term(pp + 12 * m, K, xy_all + 2 * (base + p), xyz_all + 3 * (base + p), tthr2));
    row[k] = d_out;
  }
}

// which = 0: energy of (accepted poses, lab) -> acc[0..1]; 1: of (candidate poses, lab).
// `lab_all`: labels in POSITION order; the data terms come from Work::pearl_dt (pearl_data
// with the same `which` runs before this launch).
template <bool LISTS>
__global__ __launch_bounds__(256) void pearl_energy(
    const double* __restrict__ xy_all, const double* __restrict__ xyz_all,
    const int64_t* __restrict__ slot_base, const double* __restrict__ Ks,
    const int32_t* __restrict__ num_models, EposFitParams prm, int max_k, Work w,
    const double* __restrict__ poses, const uint8_t* __restrict__ lab_all, int which) {
  const int s = blockIdx.y;
  if (!w.pearl_state[s] || (w.nb_ok[s] != 0) != LISTS) return;
  if (which == 1 && !w.pearl_moved[s]) return;
  const int lane = threadIdx.x & 63;
  const int64_t base = slot_base[s], n = slot_base[s + 1] - base;
  const double* xy = xy_all + 2 * base;
  const double* xyz = xyz_all + 3 * base;
  const uint8_t* lab = lab_all + base;
  const int k = num_models[s];
  (void)poses; (void)Ks; (void)max_k;
  const double rad = prm.neighborhood_ball_radius;
  const double s2 = prm.scaling_from_millimeters * prm.scaling_from_millimeters;
  const double r2 = rad * rad;
  const int32_t* yorder = w.yorder ? w.yorder + base : nullptr;
  const int32_t* ypos = w.ypos ? w.ypos + base : nullptr;
  NbLists nl = {nullptr, nullptr};
  if (w.nb_ok[s]) { nl.cnt = w.nb_cnt + base * NB_W; nl.pool = w.nb_pool + base * (NB_W * NB_SUB); }
  // The two sums of a slot end in ONE pair of global atomics per workgroup, spread over
  // PEARL_BINS addresses each (pearl_commit adds the bins: integers, order free), and only
  // as many workgroups take part as give every wave ~4 points: with one point and one pair
  // of atomics per wave (the grid is sized for the pool's capacity) the 130 000 same-address
  // 64-bit atomics of a C4 image serialised at ~7 ns each -- 0.93 ms per launch, 6.4 ms of
  // fitting per image (profiles/r03/kernel_stats_c4_depth1_before.csv).
  // (~2 points per wave since the neighbour walk became four round trips: the atomics are
  // two per workgroup of 8 points, spread over 64 bins)
  const int64_t want = (n + 7) / 8;
  const int wgs = static_cast<int>(want < 1 ? 1 : want < gridDim.x ? want : gridDim.x);
  if (static_cast<int>(blockIdx.x) >= wgs) return;
  const int nwaves = wgs * 4;
  unsigned long long data = 0, smooth = 0;
  for (int64_t p = blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += nwaves) {
    const int lp = lab[ypos ? ypos[p] : p];
    const int64_t dterm = w.pearl_dt[(base + p) * PEARL_DT + (lp < k ? lp : k)];
    int diff = 0, deg = 0;
    for_each_neighbour<LISTS>(xy, xyz, n, static_cast<int32_t>(p), yorder, ypos, rad, s2, r2, lane,
                       nl, [&](int64_t q, bool valid) {
                         const int lo = lab[q];
                         deg += valid;
                         diff += valid && lo != lp;
                       });
    diff = butterfly_sum_i(diff);
    deg = butterfly_sum_i(deg);
    if (lane == 0) {
      // the FRACTION of disagreeing neighbours (degree-normalised Potts)
      if (deg > 0)
        smooth += LISTS ? static_cast<unsigned long long>(static_cast<uint32_t>(GC_Q * diff) /
                                                          static_cast<uint32_t>(deg))
                        : static_cast<unsigned long long>((static_cast<int64_t>(GC_Q) * diff) / deg);
      data += static_cast<unsigned long long>(dterm);
    }
  }
  __shared__ unsigned long long s_sum[2];
  if (threadIdx.x < 2) s_sum[threadIdx.x] = 0ull;
  __syncthreads();
  if (lane == 0) {
    atomicAdd(&s_sum[0], data);
    atomicAdd(&s_sum[1], smooth);
  }
  __syncthreads();
  if (threadIdx.x < 2 && s_sum[threadIdx.x] != 0ull)
    atomicAdd(&w.pearl_acc[(s * 4 + 2 * which + threadIdx.x) * PEARL_BINS +
                           (blockIdx.x & (PEARL_BINS - 1))], s_sum[threadIdx.x]);
}

template <bool LISTS>
__global__ __launch_bounds__(256) void pearl_sweep(
    const double* __restrict__ xy_all, const double* __restrict__ xyz_all,
    const int64_t* __restrict__ slot_base, const double* __restrict__ Ks,
    const int32_t* __restrict__ num_models, EposFitParams prm, int max_k, Work w,
    const double* __restrict__ poses, const uint8_t* __restrict__ lab_in_all,
    uint8_t* __restrict__ lab_out_all, int with_energy, uint8_t* __restrict__ lab_idx_all) {
  // lab_in / lab_out: labels in POSITION order (row-sorted; what the neighbour walk indexes);
  // lab_idx_all (the last sweep of an iteration, else null): the same labels in index order
  // for pearl_refit / pearl_commit. Data terms: Work::pearl_dt (pearl_data runs before).
  // with_energy (the FIRST sweep of an iteration): the energy of (poses, lab_in) -- the
  // "before" of pearl_commit's comparison -- falls out of this pass: a point's disagreeing
  // neighbours are deg - cnt[its label] and its data term is one of the k + 1 this pass
  // evaluates anyway. Round 4: that removed two of the four pearl_energy launches per call
  // (81 us each at C4's size); same integers, same bins.
  const int s = blockIdx.y;
  if (!w.pearl_state[s] || (w.nb_ok[s] != 0) != LISTS) return;
  const int lane = threadIdx.x & 63;
  const int64_t base = slot_base[s], n = slot_base[s + 1] - base;
  const double* xy = xy_all + 2 * base;
  const double* xyz = xyz_all + 3 * base;
  const uint8_t* lab_in = lab_in_all + base;
  uint8_t* lab_out = lab_out_all + base;
  const int k = num_models[s];
  (void)poses; (void)Ks; (void)max_k;
  const double lam = prm.spatial_coherence_weight, rad = prm.neighborhood_ball_radius;
  const double s2 = prm.scaling_from_millimeters * prm.scaling_from_millimeters;
  const double r2 = rad * rad;
  const int32_t* yorder = w.yorder ? w.yorder + base : nullptr;
  const int32_t* ypos = w.ypos ? w.ypos + base : nullptr;
  NbLists nl = {nullptr, nullptr};
  if (w.nb_ok[s]) { nl.cnt = w.nb_cnt + base * NB_W; nl.pool = w.nb_pool + base * (NB_W * NB_SUB); }
  const int nwaves = gridDim.x * 4;
  unsigned long long e_data = 0, e_smooth = 0;
  for (int64_t p = blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += nwaves) {
    // Lane m (m <= k) owns label m: its data term, then (after the walk) its cost.
    const int64_t pos = ypos ? ypos[p] : p;
    const int lp = lab_in[pos];
    const int ml = lane <= k ? lane : k;
    const int64_t D = w.pearl_dt[(base + p) * PEARL_DT + ml];
    int deg = 0, mine = 0;
    if constexpr (LISTS) {
      // at most 640 neighbours: the nine label counts ride in three words, ten bits each
      // (a third of the registers and of the wave reductions; the sums are the same integers)
      static_assert(NB_W * NB_SUB < 1024 && PEARL_MAX_K + 1 <= 9, "three 10-bit fields per word");
      unsigned pk[3] = {0u, 0u, 0u};
      for_each_neighbour<true>(xy, xyz, n, static_cast<int32_t>(p), yorder, ypos, rad, s2, r2,
                               lane, nl, [&](int64_t q, bool valid) {
                                 const unsigned l = lab_in[q];
                                 const unsigned wd = l / 3u, sh = 10u * (l - 3u * wd);
                                 const unsigned one = valid && l <= PEARL_MAX_K ? 1u << sh : 0u;
#pragma unroll
                                 for (int i = 0; i < 3; ++i) pk[i] += wd == i ? one : 0u;
                               });
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        pk[i] = static_cast<unsigned>(butterfly_sum_i(static_cast<int>(pk[i])));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int c = static_cast<int>((pk[i] >> (10 * j)) & 1023u);
          deg += c;
          mine = lane == 3 * i + j ? c : mine;
        }
      }
    } else {
      int cnt[PEARL_MAX_K + 1];
#pragma unroll
      for (int m = 0; m <= PEARL_MAX_K; ++m) cnt[m] = 0;
      for_each_neighbour<false>(xy, xyz, n, static_cast<int32_t>(p), yorder, ypos, rad, s2, r2,
                                lane, nl, [&](int64_t q, bool valid) {
                                  const int l = lab_in[q];
                                  const int lo = valid ? l : -1;
#pragma unroll
                                  for (int m = 0; m <= PEARL_MAX_K; ++m) cnt[m] += lo == m;
                                });
#pragma unroll
      for (int m = 0; m <= PEARL_MAX_K; ++m) {
        cnt[m] = butterfly_sum_i(cnt[m]);
        deg += cnt[m];
        mine = lane == m ? cnt[m] : mine;
      }
    }
    // lane m: the cost of label m (same expressions as before, one label per lane)
    // (with lists a point has at most NB_W * NB_SUB = 640 neighbours: the quotient's
    // numerator stays below 2^29 and a 32-bit division gives the same integer)
    static_assert(static_cast<int64_t>(GC_Q) * NB_W * NB_SUB < (1ll << 31), "32-bit quotient");
    const int64_t sm = deg <= 0 ? 0
        : LISTS ? static_cast<int64_t>(static_cast<uint32_t>(GC_Q * (deg - mine)) /
                                       static_cast<uint32_t>(deg))
                : (static_cast<int64_t>(GC_Q) * (deg - mine)) / deg;
    const double c = (1.0 - lam) * static_cast<double>(D) + lam * static_cast<double>(sm);
    int best = 0;
    double best_c = __shfl(c, 0, 64);
    for (int m = 1; m <= k; ++m) {                      // k is uniform; first minimum wins
      const double cm = __shfl(c, m, 64);
      if (cm < best_c) { best = m; best_c = cm; }
    }
    if (lane == 0) {
      lab_out[pos] = static_cast<uint8_t>(best);
      if (lab_idx_all) lab_idx_all[base + p] = static_cast<uint8_t>(best);
    }
    if (with_energy && (lane == lp || (lane == k && lp >= k)) && lane <= k) {
      e_data += static_cast<unsigned long long>(D);
      e_smooth += static_cast<unsigned long long>(sm);
    }
  }
  if (with_energy) {                        // as pearl_energy(which = 0): acc[0], acc[1]
    __shared__ unsigned long long s_sum[2];
    if (threadIdx.x < 2) s_sum[threadIdx.x] = 0ull;
    __syncthreads();
    if (lane <= PEARL_MAX_K && (e_data | e_smooth) != 0ull) {   // the lanes that own labels
      atomicAdd(&s_sum[0], e_data);
      atomicAdd(&s_sum[1], e_smooth);
    }
    __syncthreads();
    if (threadIdx.x < 2 && s_sum[threadIdx.x] != 0ull)
      atomicAdd(&w.pearl_acc[(s * 4 + threadIdx.x) * PEARL_BINS + (blockIdx.x & (PEARL_BINS - 1))],
                s_sum[threadIdx.x]);
  }
}

__global__ __launch_bounds__(256) void pearl_refit(
    const double* __restrict__ xy_all, const double* __restrict__ xyz_all,
    const int64_t* __restrict__ slot_base, const double* __restrict__ Ks,
    const int32_t* __restrict__ num_models, EposFitParams prm, int max_k, Work w,
    const double* __restrict__ poses, const uint8_t* __restrict__ lab_all) {
  const int s = blockIdx.x;
  if (!w.pearl_state[s]) return;
  const int t = threadIdx.x;
  __shared__ LoLds s_lo;
  __shared__ int s_cnt[4];
  LoSync sy;                         // one workgroup: the sums stay in LDS
  sy.cnt = nullptr; sy.data = nullptr; sy.timeout = nullptr; sy.spin_max = 0; sy.g = 0; sy.epoch = 0;
  const int64_t base = slot_base[s], n = slot_base[s + 1] - base;
  const double* xy = xy_all + 2 * base;
  const double* xyz = xyz_all + 3 * base;
  const uint8_t* lab = lab_all + base;
  const int k = num_models[s];
  double K[9];
  for (int i = 0; i < 9; ++i) K[i] = Ks[s * 9 + i];
  const double thr2 = prm.threshold * prm.threshold;
  // one workgroup per (slot, model): the refits of a slot's models are independent (round 4;
  // one workgroup per slot walked them one after the other: 44 -> ~22 us at two models)
  const int m = blockIdx.y;
  if (m >= k) return;
  {
    double pose[12], next[12];
    for (int i = 0; i < 12; ++i) pose[i] = poses[(static_cast<int64_t>(s) * max_k + m) * 12 + i];
    int c = 0;
    for (int64_t i = t; i < n; i += 256) c += lab[i] == m;
    c = butterfly_sum_i(c);
    if ((t & 63) == 0) s_cnt[t >> 6] = c;
    __syncthreads();
    c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
    bool ok = false;
    if (c >= prm.min_point_number)                         // uniform
      ok = !lo_pass<false>(pose, K, xy, xyz, nullptr, n, thr2, t, sy, &s_lo, lab, m, nullptr,
                           nullptr, next);
    if (t < 12) w.pearl_pose[(static_cast<int64_t>(s) * PEARL_MAX_K + m) * 12 + t] =
        ok ? next[t] : pose[t];
    if (t == 0 && ok) atomicOr(&w.pearl_moved[s], 1);         // zeroed by pearl_begin
  }
}

__global__ __launch_bounds__(256) void pearl_commit(const int64_t* slot_base,
                                                    const int32_t* num_models,
                                                    EposFitParams prm, int max_k, Work w,
                                                    double* poses, const uint8_t* lab_all,
                                                    int32_t* labels_all) {
  const int s = blockIdx.x;
  if (!w.pearl_state[s]) return;
  const int t = threadIdx.x;
  const int k = num_models[s];
  const double lam = prm.spatial_coherence_weight;
  unsigned long long a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned long long v = 0;
    for (int b = 0; b < PEARL_BINS; ++b) v += w.pearl_acc[(s * 4 + q) * PEARL_BINS + b];
    a[q] = v;
  }
  const double e_before = (1.0 - lam) * static_cast<double>(static_cast<int64_t>(a[0])) +
                          lam * static_cast<double>(static_cast<int64_t>(a[1]));
  const double e_after = (1.0 - lam) * static_cast<double>(static_cast<int64_t>(a[2])) +
                         lam * static_cast<double>(static_cast<int64_t>(a[3]));
  const bool keep = w.pearl_moved[s] && e_after < e_before;   // uniform
  __syncthreads();
  if (!keep) { if (t == 0) w.pearl_state[s] = 0; return; }
  const int64_t base = slot_base[s], n = slot_base[s + 1] - base;
  for (int i = t; i < k * 12; i += 256)
    poses[static_cast<int64_t>(s) * max_k * 12 + i] =
        w.pearl_pose[static_cast<int64_t>(s) * PEARL_MAX_K * 12 + i];
  for (int64_t i = t; i < n; i += 256) {
    const int l = lab_all[base + i];
    labels_all[base + i] = l < k ? l : -1;
  }
}

inline int64_t align_up(int64_t x) { return (x + 255) / 256 * 256; }

struct Layout {
  int64_t hyp_score, hyp_pose, hyp_count, active, n_active, done, inl_bits, total;
  int64_t words_total;
  int64_t cur_pose, cur_score, cur_count, state, tries, last_new, gq, lab_a, lab_b;
  int64_t pearl_pose, pearl_acc, pearl_state, pearl_moved, lab_c, pearl_dt;
  int64_t lo_cnt, lo_data, lo_timeout;
  int64_t nb_cnt, nb_pool, nb_ok;
  int64_t geo, dyn_a, dyn_b, win, acc, flip_a, flip_b;
};

// cooperating launches per call: select + refit of every round (max_k + 2 rounds at most)
int lo_launches(int max_k) { return 2 * (max_k + 2); }

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
  L.cur_pose = off; off = align_up(off + (S + 1) * 12 * 8);
  L.cur_score = off; off = align_up(off + (S + 1) * 8);
  L.cur_count = off; off = align_up(off + (S + 1) * 4);
  L.state = off; off = align_up(off + (S + 1) * 4);
  L.tries = off; off = align_up(off + (S + 1) * 4);
  L.last_new = off; off = align_up(off + (S + 1) * 4);
  L.gq = off; off = align_up(off + (n_cap + 1) * 4);
  L.lab_a = off; off = align_up(off + n_cap + 1);
  L.lab_b = off; off = align_up(off + n_cap + 1);
  L.lo_cnt = off; off = align_up(off + static_cast<int64_t>(S + 1) * lo_launches(max_k) * 4);
  L.lo_data = off; off = align_up(off + static_cast<int64_t>(S + 1) * 2 * LO_G * 4 * LO_NV * 8);
  L.lo_timeout = off; off = align_up(off + 8);
  L.nb_cnt = off; off = align_up(off + (n_cap + 1) * NB_W * 2);
  L.nb_pool = off; off = align_up(off + (n_cap + 1) * NB_W * NB_SUB * 2);
  L.nb_ok = off; off = align_up(off + (S + 1) * 4);
  L.geo = off; off = align_up(off + (n_cap + 1) * 32);
  L.dyn_a = off; off = align_up(off + (n_cap + 1) * 16);
  L.dyn_b = off; off = align_up(off + (n_cap + 1) * 16);
  L.win = off; off = align_up(off + L.words_total * 2 * 4);
  L.acc = off; off = align_up(off + (n_cap + 1) * 16);
  L.flip_a = off; off = align_up(off + n_cap + 1);
  L.flip_b = off; off = align_up(off + n_cap + 1);
  L.pearl_pose = off; off = align_up(off + (S + 1) * PEARL_MAX_K * 12 * 8);
  L.pearl_acc = off; off = align_up(off + (S + 1) * 4 * PEARL_BINS * 8);
  L.pearl_state = off; off = align_up(off + (S + 1) * 4);
  L.pearl_moved = off; off = align_up(off + (S + 1) * 4);
  L.lab_c = off; off = align_up(off + n_cap + 1);
  L.pearl_dt = off; off = align_up(off + (n_cap + 1) * PEARL_DT * 4);
  L.total = off;
  return L;
}

// Enqueues the whole fitting stage. yorder / ypos [device, n_capacity] or null: the
// y-sorted order of every slot's correspondences and its inverse (slot-local indices);
// null = the slots are already sorted by image row (epos_corr_fill's raster order).
int find6d_enqueue(const double* xy, const double* xyz, const int64_t* slot_base, int S,
                   int64_t n_capacity, const double* Ks, const int32_t* max_models,
                   const uint64_t* seeds, const EposFitParams* p, int32_t max_k, void* work,
                   double* poses, double* scores, int32_t* num_models, int32_t* labels,
                   const int32_t* yorder, const int32_t* ypos, hipStream_t st) {
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
  w.cur_pose = reinterpret_cast<double*>(wb + L.cur_pose);
  w.cur_score = reinterpret_cast<double*>(wb + L.cur_score);
  w.cur_count = reinterpret_cast<int32_t*>(wb + L.cur_count);
  w.state = reinterpret_cast<int32_t*>(wb + L.state);
  w.tries = reinterpret_cast<int32_t*>(wb + L.tries);
  w.last_new = reinterpret_cast<int32_t*>(wb + L.last_new);
  w.gq = reinterpret_cast<int32_t*>(wb + L.gq);
  w.lab_a = reinterpret_cast<uint8_t*>(wb + L.lab_a);
  w.lab_b = reinterpret_cast<uint8_t*>(wb + L.lab_b);
  w.lab_c = reinterpret_cast<uint8_t*>(wb + L.lab_c);
  w.pearl_dt = reinterpret_cast<int32_t*>(wb + L.pearl_dt);
  w.yorder = yorder;
  w.ypos = ypos;
  w.lo_cnt = reinterpret_cast<unsigned*>(wb + L.lo_cnt);
  w.lo_data = reinterpret_cast<double*>(wb + L.lo_data);
  w.lo_timeout = reinterpret_cast<int32_t*>(wb + L.lo_timeout);
  static const unsigned spin_max = [] {      // EPOS_FIT_SPIN_MAX=0: every hand-off that has to
    const char* e = getenv("EPOS_FIT_SPIN_MAX");   // wait at all gives up (tests of the error path)
    return e ? static_cast<unsigned>(strtoul(e, nullptr, 10)) : LO_SPIN_MAX;
  }();
  w.lo_spin_max = spin_max;
  w.nb_cnt = reinterpret_cast<uint16_t*>(wb + L.nb_cnt);
  w.nb_pool = reinterpret_cast<int16_t*>(wb + L.nb_pool);
  w.nb_ok = reinterpret_cast<int32_t*>(wb + L.nb_ok);
  w.geo = reinterpret_cast<double*>(wb + L.geo);
  w.dyn_a = reinterpret_cast<GcDyn*>(wb + L.dyn_a);
  w.dyn_b = reinterpret_cast<GcDyn*>(wb + L.dyn_b);
  w.win = reinterpret_cast<int32_t*>(wb + L.win);
  w.acc = reinterpret_cast<GcAcc*>(wb + L.acc);
  w.flip_a = reinterpret_cast<int8_t*>(wb + L.flip_a);
  w.flip_b = reinterpret_cast<int8_t*>(wb + L.flip_b);
  w.pearl_pose = reinterpret_cast<double*>(wb + L.pearl_pose);
  w.pearl_acc = reinterpret_cast<unsigned long long*>(wb + L.pearl_acc);
  w.pearl_state = reinterpret_cast<int32_t*>(wb + L.pearl_state);
  w.pearl_moved = reinterpret_cast<int32_t*>(wb + L.pearl_moved);
  const int n_lo = lo_launches(max_k);
  const bool gc = p->gc_sweeps > 0 && p->spatial_coherence_weight > 0.0 &&
                  p->neighborhood_ball_radius > 0.0;
  static const int use_nb = [] {            // EPOS_FIT_NB_LISTS=0: scan windows every sweep
    const char* e = getenv("EPOS_FIT_NB_LISTS");
    return e ? atoi(e) : 1;
  }();
  // building the lists costs one sweep's worth of pair tests (~100 us for 5 x 5 000
  // correspondences), walking them less than half a sweep: it pays from the third
  // neighbourhood pass of a call on (multi-instance searches, the joint refinement); the
  // single-instance call of C2 (one round, two sweeps) breaks even and keeps the scans
  // Round 3, later: with the candidates in scalar registers (ransac_gc_scan) a window scan
  // costs less than walking the lists, so the binary sweeps always scan and the lists are
  // only built for the joint refinement, whose passes gather per point (EPOS_FIT_SCAN=0:
  // the LDS sweeps and the rule above).
  static const int use_scan = [] {
    const char* e = getenv("EPOS_FIT_SCAN");
    return e ? atoi(e) : 1;
  }();
  static const int use_delta = [] {          // EPOS_FIT_DELTA=0: every sweep is a full scan
    const char* e = getenv("EPOS_FIT_DELTA");
    return e ? atoi(e) : 1;
  }();
  const int nb_rounds = max_k + (max_k > 1 ? 2 : 0);
  const bool nb_pays = use_scan ? (max_k >= 2 && p->pearl_iters > 0)
                                : nb_rounds * p->gc_sweeps > 2;
  const int build_nb = gc && use_nb && GC_W == NB_W &&     // lists are laid out per wave
                       nb_pays ? 1 : 0;
  hipLaunchKernelGGL(ransac_init, dim3(S, 24), dim3(256), 0, st, xy, xyz, slot_base, S, w, labels,
                     num_models, p->min_point_number, n_capacity, n_lo, build_nb, gc ? 1 : 0,
                     p->neighborhood_ball_radius);
  int rc = launch_status("ransac_init");
  if (rc) return rc;
  const dim3 hgrid(static_cast<unsigned>(ceil_div(p->max_iters, 4)), S);
  // a multi-instance (Progressive-X) search may retry a failed proposal: two extra rounds
  const int rounds = max_k + (max_k > 1 ? 2 : 0);
  // joint refinement kernels: one wavefront per point and pass
  const int64_t per_slot = ceil_div(n_capacity, S > 0 ? S : 1);
  // (512 .. 2048 workgroups per slot measure the same at C4's size; 128 is 30 % slower)
  const dim3 pgrid(static_cast<unsigned>(per_slot < 8192 ? ceil_div(per_slot, 4) < 64 ? 64 : ceil_div(per_slot, 4) : 2048), S);
  // labelling sweeps: one workgroup per tile of 64 points (grid-stride beyond 512 tiles)
  const int64_t tiles = ceil_div(per_slot, 64);
  const dim3 sgrid(static_cast<unsigned>(tiles < 1 ? 1 : tiles > 512 ? 512 : tiles), S);
  if (build_nb) {       // the neighbourhood graph, once: neither labels nor rounds change it
    hipLaunchKernelGGL(ransac_gc_sweep<true>, sgrid, dim3(GC_T), 0, st, xy, xyz, slot_base, *p,
                       w, nullptr, nullptr);
    rc = launch_status("ransac_nb_build");
    if (rc) return rc;
  }
  for (int round = 0; round < rounds; ++round) {
    hipLaunchKernelGGL(ransac_hypotheses, hgrid, dim3(256), 0, st, xy, xyz,
                       slot_base, Ks, seeds, max_models, num_models, *p, max_k,
                       round, w);
    rc = launch_status("ransac_hypotheses");
    if (rc) return rc;
    hipLaunchKernelGGL(ransac_select_lo, dim3(LO_G, S), dim3(256), 0, st, xy, xyz, slot_base,
                       Ks, max_models, *p, max_k, round, w, num_models, labels, 2 * round,
                       n_lo);
    rc = launch_status("ransac_select_lo");
    if (rc) return rc;
    const uint8_t* lab_final = w.lab_a;
    if (gc) {
      for (int sw = 0; sw < p->gc_sweeps; ++sw) {
        const uint8_t* in = (sw & 1) ? w.lab_b : w.lab_a;
        uint8_t* out = (sw & 1) ? w.lab_a : w.lab_b;
        if (use_scan && sw > 0 && use_delta)
          hipLaunchKernelGGL(ransac_gc_delta, sgrid, dim3(256), 0, st, slot_base, *p, w,
                             reinterpret_cast<const GcGeo*>(w.geo), w.dyn_a,
                             (sw & 1) ? w.flip_a : w.flip_b, (sw & 1) ? w.flip_b : w.flip_a,
                             in, out);
        else if (use_scan)
          hipLaunchKernelGGL(ransac_gc_scan,
                             dim3(static_cast<unsigned>(ceil_div(sgrid.x, GS_P)), S), dim3(GS_T),
                             0, st, slot_base, *p, w,
                             reinterpret_cast<const GcGeo*>(w.geo),
                             (sw & 1) ? w.dyn_b : w.dyn_a, (sw & 1) ? w.dyn_a : w.dyn_b, in,
                             out, (sw & 1) ? w.flip_b : w.flip_a);
        else
          hipLaunchKernelGGL(ransac_gc_sweep<false>, sgrid, dim3(GC_T), 0, st, xy, xyz,
                             slot_base, *p, w, in, out);
        rc = launch_status("ransac_gc_sweep");
        if (rc) return rc;
        lab_final = out;
      }
    }
    hipLaunchKernelGGL(ransac_refit_accept, dim3(LO_G, S), dim3(256), 0, st, xy, xyz,
                       slot_base, Ks, max_models, *p, max_k, w, lab_final, poses, scores,
                       num_models, labels, 2 * round + 1, n_lo);
    rc = launch_status("ransac_refit_accept");
    if (rc) return rc;
  }
  // joint refinement of multi-instance slots
  if (max_k >= 2 && p->pearl_iters > 0 && gc) {
    hipLaunchKernelGGL(pearl_setup, dim3(static_cast<unsigned>(ceil_div(S, 64))), dim3(64), 0, st,
                       num_models, *p, w, S);
    for (int it = 0; it < p->pearl_iters; ++it) {
      hipLaunchKernelGGL(pearl_begin, pgrid, dim3(256), 0, st, slot_base, num_models, w, labels);
      // the energy of (accepted poses, current labels) comes out of the first sweep
      // (pearl_setup admits a slot only with gc_sweeps >= 1)
      // (sweep labels live in position order: lab_a / lab_b; the last sweep also writes them
      // in index order into lab_c for the refit and the commit)
      const dim3 dgrid(static_cast<unsigned>(pgrid.x < 4 ? 1 : pgrid.x / 4), S);
      const dim3 fgrid(pgrid.x < 256 ? pgrid.x : 256, S);
      hipLaunchKernelGGL(pearl_data, dgrid, dim3(256), 0, st, xy, xyz, slot_base, Ks, num_models,
                         *p, max_k, w, poses, 0);
      const uint8_t* lab_final = w.lab_a;
      for (int sw = 0; sw < p->gc_sweeps; ++sw) {
        const uint8_t* in = (sw & 1) ? w.lab_b : w.lab_a;
        uint8_t* out = (sw & 1) ? w.lab_a : w.lab_b;
        // (one launch per case of Work::nb_ok; the slots of the other case leave at once.
        // The window-walk case is the rare one -- a sub-list overflowed -- and gets the
        // small grid: its launch is ~2 us of empty workgroups otherwise)
        hipLaunchKernelGGL(pearl_sweep<true>, pgrid, dim3(256), 0, st, xy, xyz, slot_base, Ks,
                           num_models, *p, max_k, w, poses, in, out, sw == 0 ? 1 : 0,
                           sw + 1 == p->gc_sweeps ? w.lab_c : nullptr);
        hipLaunchKernelGGL(pearl_sweep<false>, fgrid, dim3(256), 0, st, xy, xyz, slot_base, Ks,
                           num_models, *p, max_k, w, poses, in, out, sw == 0 ? 1 : 0,
                           sw + 1 == p->gc_sweeps ? w.lab_c : nullptr);
        lab_final = out;
      }
      hipLaunchKernelGGL(pearl_refit, dim3(S, PEARL_MAX_K), dim3(256), 0, st, xy, xyz, slot_base, Ks,
                         num_models, *p, max_k, w, poses, w.lab_c);
      hipLaunchKernelGGL(pearl_data, dgrid, dim3(256), 0, st, xy, xyz, slot_base, Ks, num_models,
                         *p, max_k, w, poses, 1);
      hipLaunchKernelGGL(pearl_energy<true>, pgrid, dim3(256), 0, st, xy, xyz, slot_base, Ks,
                         num_models, *p, max_k, w, w.pearl_pose, lab_final, 1);
      hipLaunchKernelGGL(pearl_energy<false>, fgrid, dim3(256), 0, st, xy, xyz, slot_base, Ks,
                         num_models, *p, max_k, w, w.pearl_pose, lab_final, 1);
      hipLaunchKernelGGL(pearl_commit, dim3(S), dim3(256), 0, st, slot_base, num_models, *p,
                         max_k, w, poses, w.lab_c, labels);
      rc = launch_status("pearl");
      if (rc) return rc;
    }
  }
  return EPOS_OK;
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
  p->gc_sweeps = 2;
  p->pearl_iters = 2;
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
  EPOS_REQUIRE(p->gc_sweeps >= 0 && p->gc_sweeps <= 16, "gc_sweeps must be in [0, 16]");
  EPOS_REQUIRE(p->pearl_iters >= 0 && p->pearl_iters <= 8, "pearl_iters must be in [0, 8]");
  if (S == 0) return EPOS_OK;
  return find6d_enqueue(xy, xyz, slot_base, S, n_capacity, Ks, max_models, seeds, p, max_k,
                        work, poses, scores, num_models, labels, nullptr, nullptr,
                        static_cast<hipStream_t>(stream));
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
                           static_cast<int64_t>(max_k) * 8, 8, n * 4, n * 4, n * 4};
  constexpr int ND = 13;
  char* d[ND] = {0};
  int rc = EPOS_OK;
  for (int i = 0; i < ND && !rc; ++i)
    rc = check_hip(hipMalloc(reinterpret_cast<void**>(&d[i]), sizes[i] + 8), "hipMalloc");
  int32_t k = 0;
  // The caller's order is kept (PROSAC samples from a prefix of it); the neighbourhood
  // windows of the spatial-coherence step walk the correspondences in image-row order
  // through this permutation (stable sort by y) and its inverse.
  int32_t* yorder = static_cast<int32_t*>(malloc(sizeof(int32_t) * static_cast<size_t>(n) * 2));
  if (!yorder) { set_error("epos_find6d_poses: out of host memory"); rc = EPOS_E_INVALID; }
  if (!rc) {
    int32_t* ypos = yorder + n;
    for (int64_t i = 0; i < n; ++i) yorder[i] = static_cast<int32_t>(i);
    std::stable_sort(yorder, yorder + n, [&](int32_t a, int32_t b) {
      return xy[2 * static_cast<int64_t>(a) + 1] < xy[2 * static_cast<int64_t>(b) + 1];
    });
    for (int64_t i = 0; i < n; ++i) ypos[yorder[i]] = static_cast<int32_t>(i);
    rc = check_hip(hipMemcpy(d[11], yorder, n * 4, hipMemcpyHostToDevice), "copy yorder");
    if (!rc) rc = check_hip(hipMemcpy(d[12], ypos, n * 4, hipMemcpyHostToDevice), "copy ypos");
  }
  free(yorder);
  if (!rc) {
    const int64_t sb[2] = {0, n};
    const int32_t mm = want;
    rc = check_hip(hipMemcpy(d[0], xy, n * 16, hipMemcpyHostToDevice), "copy xy");
    if (!rc) rc = check_hip(hipMemcpy(d[1], xyz, n * 24, hipMemcpyHostToDevice), "copy xyz");
    if (!rc) rc = check_hip(hipMemcpy(d[2], sb, 16, hipMemcpyHostToDevice), "copy base");
    if (!rc) rc = check_hip(hipMemcpy(d[3], K, 72, hipMemcpyHostToDevice), "copy K");
    if (!rc) rc = check_hip(hipMemcpy(d[4], &mm, 4, hipMemcpyHostToDevice), "copy mm");
    if (!rc) rc = check_hip(hipMemcpy(d[5], &seed, 8, hipMemcpyHostToDevice), "copy seed");
    if (!rc && (p->max_iters < 1 || p->gc_sweeps < 0 || p->gc_sweeps > 16 ||
                p->pearl_iters < 0 || p->pearl_iters > 8)) {
      set_error("epos_find6d_poses: max_iters >= 1, gc_sweeps in [0, 16], pearl_iters in [0, 8]");
      rc = EPOS_E_INVALID;
    }
    if (!rc)
      rc = find6d_enqueue(
          reinterpret_cast<double*>(d[0]), reinterpret_cast<double*>(d[1]),
          reinterpret_cast<int64_t*>(d[2]), 1, n, reinterpret_cast<double*>(d[3]),
          reinterpret_cast<int32_t*>(d[4]), reinterpret_cast<uint64_t*>(d[5]), p,
          max_k, d[6], reinterpret_cast<double*>(d[7]),
          reinterpret_cast<double*>(d[8]), reinterpret_cast<int32_t*>(d[9]),
          reinterpret_cast<int32_t*>(d[10]), reinterpret_cast<int32_t*>(d[11]),
          reinterpret_cast<int32_t*>(d[12]), nullptr);
    if (!rc) rc = check_hip(hipDeviceSynchronize(), "sync");
    if (!rc) rc = check_hip(hipMemcpy(&k, d[9], 4, hipMemcpyDeviceToHost), "copy k");
    if (!rc && k < 0) {
      set_error("epos_find6d_poses: a hand-off between cooperating workgroups timed out");
      rc = EPOS_E_INTERNAL;
    }
    if (!rc && k > 0) {
      rc = check_hip(hipMemcpy(poses_out, d[7], static_cast<size_t>(k) * 96, hipMemcpyDeviceToHost), "copy poses");
      if (!rc) rc = check_hip(hipMemcpy(scores_out, d[8], static_cast<size_t>(k) * 8, hipMemcpyDeviceToHost), "copy scores");
    }
    if (!rc) rc = check_hip(hipMemcpy(labels_out, d[10], n * 4, hipMemcpyDeviceToHost), "copy labels");
  }
  for (int i = 0; i < ND; ++i)
    if (d[i]) (void)hipFree(d[i]);
  return rc ? rc : k;
}

#ifdef EPOS_FIT_TRACE
extern "C" int epos_debug_fit_trace(unsigned long long* out /*[8][32]*/, int* n8, int reset) {
  int rc = static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(epos::g_fit_trace), 8 * 32 * 8));
  rc |= static_cast<int>(hipMemcpyFromSymbol(n8, HIP_SYMBOL(epos::g_fit_trace_n), 32));
  if (reset) {
    const int z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    rc |= static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(epos::g_fit_trace_n), z, 32));
  }
  return rc;
}
#endif

#ifdef EPOS_GC_STATS
extern "C" int epos_debug_gc_stats(unsigned long long* out4, int reset) {
  int rc = static_cast<int>(hipMemcpyFromSymbol(out4, HIP_SYMBOL(epos::g_gc_stats), 32));
  if (reset) {
    const unsigned long long z[4] = {0, 0, 0, 0};
    rc |= static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(epos::g_gc_stats), z, 32));
  }
  return rc;
}
#endif
