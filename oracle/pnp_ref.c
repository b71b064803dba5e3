/*
 * ORACLE (test infrastructure, not product): plain-C, single-thread restatement of
 * the EPOS pose-fitting stage, i.e. of the call
 *     pyprogressivex.find6DPoses(...)                 scripts/infer.py:470-488
 * and of how its result is unpacked (infer.py:490-503).
 *
 * PARITY UNPINNED. The arithmetic lives in the un-vendored submodule
 * danini/progressive-x (branch version-epos, .gitmodules:7-10; the commit is not
 * recorded in the tree, the directory is empty) and the reference holds no test
 * or golden pose for it. What is fixed by the reference is the call contract
 * only: f64 2D/3D correspondences, K, the parameter set, ">= 6 correspondences",
 * "up to num_instances poses, each [R|t] + a quality, or None". This file
 * restates the PUBLISHED algorithm family (Barath & Matas, GC-RANSAC CVPR'18 /
 * Progressive-X ICCV'19; EPOS CVPR'20 sec. 3.4) in the form the build defines
 * (DESIGN.md "Pose fitting"):
 *
 *   per proposal round (GC-RANSAC as the proposal engine of Progressive-X):
 *     for it in [0, max_iters):
 *        draw 3 distinct active correspondences
 *        P3P minimal solver -> <= 4 poses
 *        MSAC quality  q = sum_inliers (1 - e^2 / tau_r^2),  e = reprojection error
 *        keep the best so far (strictly greater wins); stop as soon as
 *        (1 - w^3)^(it+1) <= 1 - proposal_engine_conf, w = its inlier ratio (the
 *        RANSAC bound; with the EPOS default 1.0 the cap always runs)
 *     local optimisation of the best hypothesis:
 *        (i)  Gauss-Newton refits of the 6-dof pose on its inliers, kept while q grows
 *        (ii) spatial-coherence labelling (GC-RANSAC's energy on the neighbourhood
 *             graph, see gc_label below) -> refits on the LABELLED inliers, kept while q
 *             grows
 *     accept iff  #inliers >= min_point_number,
 *                 Tanimoto(inliers, inliers of each accepted instance) < max_tanimoto,
 *                 coverage = |inliers not yet explained| / |inliers| >= min_coverage
 *     label + remove its inliers; stop at max_model_number instances.
 *     A failed proposal ends a single-instance search (max_model_number == 1: "only
 *     GC-RANSAC is applied", infer.py:456-459). In the multi-instance search it is
 *     retried with fresh samples while the samples drawn since the last success have
 *     not yet reached confidence `conf` of having hit an instance as large as the last
 *     accepted one -- Progressive-X's termination criterion -- within a budget of two
 *     extra rounds.
 *   final joint optimisation (PEARL's role) when 2 <= #instances <=
 *     max_model_number_for_optimization: points are re-assigned to the instance that
 *     explains them best (residual + neighbourhood agreement), every instance is refitted
 *     on its points; kept when the joint energy drops (pearl_refine below).
 *
 * What is an APPROXIMATION of the published method here, on purpose (documented in
 * DESIGN.md): the binary labelling minimises GC-RANSAC's energy by synchronous
 * iterated conditional modes (gc_sweeps Jacobi sweeps from the thresholded labelling)
 * instead of an exact s-t min-cut, with energies in 2^-20 fixed point so that any
 * evaluation order gives the same labels; one labelling per proposal instead of one per
 * local-optimisation step; PEARL's alpha-expansion is likewise replaced by sweeps.
 * Against the exact s-t minimum cut of the same energy (tests/test_oracle_fit.py): two sweeps
 * ARE the minimum cut on sparse neighbourhood graphs (<= 5 neighbours per point), within
 * 0-1.6 % of its energy at 10-14 neighbours; beyond 2 (1 - lambda) / lambda = 18 neighbours
 * the exact minimiser labels (nearly) everything an inlier and the sweeps are a bounded
 * step towards it. gc_sweeps < 0 selects that exact minimiser here
 * (gc_label_exact: max-flow; a checker-only mode the HIP product does not have), so that the
 * tests can assert where sweeps and cut coincide.
 *
 * All sums over correspondences use canonical orders so that the wavefront-
 * parallel HIP kernels can reproduce them bit for bit:
 *   - hypothesis scoring (one wavefront per hypothesis): 64 strided partial sums
 *     (partial l takes items l, l+64, ...) combined by a 6-level xor butterfly;
 *   - local optimisation of a proposal (round 3: FOUR 256-thread workgroups per object):
 *     1024 strided partial sums (partial q takes items q, q+1024, ...), a 6-level xor
 *     butterfly inside each group of 64, (g0 + g1) + (g2 + g3) inside each group of 256,
 *     then (W0 + W1) + (W2 + W3); the joint refinement (one workgroup per object) keeps
 *     256 partials and stops after the second level.
 * The projection uses ONE reciprocal per point (iz = 1/Z, then multiplications) and
 * the MSAC term is 1 - e2 * (1/thr2): a division costs the GPU about ten fp64
 * instructions. Only + - * / sqrt are used (correctly rounded on both sides);
 * build with -ffp-contract=off.
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct PnpRefParams {
  double threshold;
  double neighborhood_ball_radius;
  double spatial_coherence_weight;
  double scaling_from_millimeters;
  double max_tanimoto_similarity;
  double conf;
  double proposal_engine_conf;
  double min_coverage;
  double min_triangle_area;
  int32_t max_iters;
  int32_t min_point_number;
  int32_t max_model_number;
  int32_t max_model_number_for_optimization;
  int32_t use_prosac;
  int32_t lo_iters;
  int32_t gc_sweeps;     /* relabelling sweeps of the spatial-coherence step (0 = off;
                          * < 0 = exact s-t minimum cut, oracle only: gc_label_exact) */
  int32_t pearl_iters;   /* joint refinement iterations (0 = off) */
} PnpRefParams;

/* ------------------------------------------------------------------ RNG -- */
static uint64_t mix64(uint64_t z) {          /* splitmix64 finaliser */
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static uint64_t draw(uint64_t seed, uint32_t round, uint32_t it, uint32_t k,
                     uint64_t n) {           /* uniform in [0, n) */
  uint64_t u = mix64(mix64(seed ^ mix64(((uint64_t)round << 32) | it)) + k);
  return (uint64_t)(((unsigned __int128)u * n) >> 64);
}
/* three distinct indices in [0, m) */
static void sample3(uint64_t seed, uint32_t round, uint32_t it, int64_t m,
                    int64_t s[3]) {
  int64_t a = (int64_t)draw(seed, round, it, 0, (uint64_t)m);
  int64_t b = (int64_t)draw(seed, round, it, 1, (uint64_t)(m - 1));
  int64_t c = (int64_t)draw(seed, round, it, 2, (uint64_t)(m - 2));
  if (b >= a) b += 1;
  int64_t lo = a < b ? a : b, hi = a < b ? b : a;
  if (c >= lo) c += 1;
  if (c >= hi) c += 1;
  s[0] = a; s[1] = b; s[2] = c;
}

/* ------------------------------------------------------- small algebra -- */
static double dot3(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
/* symmetric 3x3 stored as s[6] = {00, 01, 02, 11, 12, 22} */
static void sym_adj(const double* s, double* b) {
  b[0] = s[3] * s[5] - s[4] * s[4];
  b[1] = s[2] * s[4] - s[1] * s[5];
  b[2] = s[1] * s[4] - s[2] * s[3];
  b[3] = s[0] * s[5] - s[2] * s[2];
  b[4] = s[1] * s[2] - s[0] * s[4];
  b[5] = s[0] * s[3] - s[1] * s[1];
}
static double sym_det(const double* s, const double* adj) {
  return s[0] * adj[0] + s[1] * adj[1] + s[2] * adj[2];
}
static double sym_inner(const double* a, const double* b) { /* sum_ij a_ij b_ij */
  return a[0] * b[0] + a[3] * b[3] + a[5] * b[5] +
         2.0 * (a[1] * b[1] + a[2] * b[2] + a[4] * b[4]);
}
static double sym_quad(const double* s, const double* v) { /* v^T S v */
  return s[0] * v[0] * v[0] + s[3] * v[1] * v[1] + s[5] * v[2] * v[2] +
         2.0 * (s[1] * v[0] * v[1] + s[2] * v[0] * v[2] + s[4] * v[1] * v[2]);
}
static double sym_at(const double* s, int i, int j) {
  static const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  return s[idx[i][j]];
}

/* One real root of x^3 + b x^2 + c x + d (Newton from a start beyond the outer
 * turning point, so the iteration is monotone). */
static double cubic_root(double b, double c, double d) {
  double x;
  const double disc = b * b - 3.0 * c;
  if (disc > 0.0) {
    const double v = sqrt(disc);
    const double t1 = (-b - v) / 3.0;            /* local maximum */
    const double f1 = ((t1 + b) * t1 + c) * t1 + d;
    if (f1 > 0.0) {                              /* root left of the maximum */
      double step = v / 3.0 + 1e-3;
      x = t1 - step;
      for (int i = 0; i < 200 && (((x + b) * x + c) * x + d) > 0.0; ++i) { step *= 2.0; x = t1 - step; }
    } else {
      const double t2 = (-b + v) / 3.0;          /* local minimum */
      double step = v / 3.0 + 1e-3;
      x = t2 + step;
      for (int i = 0; i < 200 && (((x + b) * x + c) * x + d) < 0.0; ++i) { step *= 2.0; x = t2 + step; }
    }
  } else {
    /* monotone cubic: bracket from the inflection point outward */
    const double t0 = -b / 3.0;
    const double f0 = ((t0 + b) * t0 + c) * t0 + d;
    double step = 1.0 + fabs(t0);
    x = t0;
    if (f0 > 0.0) {
      x = t0 - step;
      for (int i = 0; i < 200 && (((x + b) * x + c) * x + d) > 0.0; ++i) { step *= 2.0; x = t0 - step; }
    } else if (f0 < 0.0) {
      x = t0 + step;
      for (int i = 0; i < 200 && (((x + b) * x + c) * x + d) < 0.0; ++i) { step *= 2.0; x = t0 + step; }
    }
  }
  for (int i = 0; i < 60; ++i) {
    const double f = ((x + b) * x + c) * x + d;
    const double fp = (3.0 * x + 2.0 * b) * x + c;
    if (fp == 0.0) break;
    const double dx = f / fp;
    x -= dx;
    if (fabs(dx) <= 1e-15 * fabs(x)) break;
  }
  return x;
}

/* ------------------------------------------------------------------ P3P -- */
/* f[3][3]: unit bearings, X[3][3]: world points. Out: up to 4 poses as R (row
 * major 9) + t (3) in pose[k*12..]. Returns the number of poses. */
static int p3p(const double f[3][3], const double X[3][3], double* pose) {
  const double c12 = dot3(f[0], f[1]), c13 = dot3(f[0], f[2]),
               c23 = dot3(f[1], f[2]);
  double d12[3], d13[3], d23[3];
  for (int i = 0; i < 3; ++i) {
    d12[i] = X[0][i] - X[1][i];
    d13[i] = X[0][i] - X[2][i];
    d23[i] = X[1][i] - X[2][i];
  }
  const double a12 = dot3(d12, d12), a13 = dot3(d13, d13), a23 = dot3(d23, d23);
  /* world-triangle frame and its inverse */
  double nx[3];
  cross3(d12, d13, nx);
  const double detx = dot3(nx, nx);
  if (!(detx > 1e-18 * (a12 * a13 + 1e-300))) return 0;   /* collinear sample */
  /* Xm = [d12 d13 nx] (columns); inverse rows: (d13 x nx, nx x d12, nx) / detx */
  double r0[3], r1[3];
  cross3(d13, nx, r0);
  cross3(nx, d12, r1);
  double Xinv[3][3];
  for (int i = 0; i < 3; ++i) {
    Xinv[0][i] = r0[i] / detx;
    Xinv[1][i] = r1[i] / detx;
    Xinv[2][i] = nx[i] / detx;
  }
  /* quadrics: M12, M13, M23 in packed symmetric form */
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
  /* det(D1 + g D2) = k0 + k1 g + k2 g^2 + k3 g^3 */
  double D0[6];
  const double* E;                 /* the conic the line pair is intersected with */
  if (fabs(k3) >= fabs(k0)) {
    if (k3 == 0.0) return 0;
    const double g = cubic_root(k2 / k3, k1 / k3, k0 / k3);
    for (int i = 0; i < 6; ++i) D0[i] = D1[i] + g * D2[i];
    E = fabs(g) <= 1.0 ? D2 : D1;
  } else {
    const double g = cubic_root(k1 / k0, k2 / k0, k3 / k0);   /* D0 = g D1 + D2 */
    for (int i = 0; i < 6; ++i) D0[i] = g * D1[i] + D2[i];
    E = fabs(g) <= 1.0 ? D1 : D2;
  }
  /* split the degenerate conic D0 into two planes l, m */
  double B[6];
  sym_adj(D0, B);
  int bi = 0;
  double bmax = -B[0];
  if (-B[3] > bmax) { bmax = -B[3]; bi = 1; }
  if (-B[5] > bmax) { bmax = -B[5]; bi = 2; }
  if (!(bmax > 0.0)) return 0;                   /* complex line pair */
  const double sq = sqrt(bmax);
  double pt[3];
  for (int i = 0; i < 3; ++i) pt[i] = -sym_at(B, i, bi) / sq;
  double N[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) N[i][j] = sym_at(D0, i, j);
  N[0][1] -= pt[2]; N[0][2] += pt[1];
  N[1][0] += pt[2]; N[1][2] -= pt[0];
  N[2][0] -= pt[1]; N[2][1] += pt[0];
  /* rank one: N = 2 m l^T; strongest row ~ l, strongest column ~ m */
  int ri = 0, ci = 0;
  double rbest = -1.0, cbest = -1.0;
  for (int i = 0; i < 3; ++i) {
    const double rn = N[i][0] * N[i][0] + N[i][1] * N[i][1] + N[i][2] * N[i][2];
    const double cn = N[0][i] * N[0][i] + N[1][i] * N[1][i] + N[2][i] * N[2][i];
    if (rn > rbest) { rbest = rn; ri = i; }
    if (cn > cbest) { cbest = cn; ci = i; }
  }
  double planes[2][3];
  for (int j = 0; j < 3; ++j) { planes[0][j] = N[ri][j]; planes[1][j] = N[j][ci]; }

  int nsol = 0;
  for (int pl = 0; pl < 2; ++pl) {
    const double* n = planes[pl];
    int k = 0;
    if (fabs(n[1]) > fabs(n[k])) k = 1;
    if (fabs(n[2]) > fabs(n[k])) k = 2;
    if (n[k] == 0.0) continue;
    const int a = (k + 1) % 3, b = (k + 2) % 3;
    const double ua = -n[a] / n[k], ub = -n[b] / n[k];   /* lam_k = ua la + ub lb */
    /* q = P^T E P with P columns e_a + ua e_k, e_b + ub e_k */
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
      lam[a] = la; lam[b] = lb; lam[k] = ua * la + ub * lb;
      const double qs = sym_quad(M12, lam) + sym_quad(M13, lam) + sym_quad(M23, lam);
      if (!(qs > 0.0)) continue;
      double sc = sqrt((a12 + a13 + a23) / qs);
      if (lam[0] < 0.0) sc = -sc;
      lam[0] *= sc; lam[1] *= sc; lam[2] *= sc;
      if (!(lam[0] > 0.0 && lam[1] > 0.0 && lam[2] > 0.0)) continue;
      /* camera-frame triangle */
      double Y[3][3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Y[i][j] = lam[i] * f[i][j];
      double e12[3], e13[3], ny[3];
      for (int j = 0; j < 3; ++j) { e12[j] = Y[0][j] - Y[1][j]; e13[j] = Y[0][j] - Y[2][j]; }
      cross3(e12, e13, ny);
      double* R = pose + nsol * 12;
      double* t = R + 9;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          R[i * 3 + j] = e12[i] * Xinv[0][j] + e13[i] * Xinv[1][j] + ny[i] * Xinv[2][j];
      for (int i = 0; i < 3; ++i)
        t[i] = Y[0][i] - (R[i * 3] * X[0][0] + R[i * 3 + 1] * X[0][1] + R[i * 3 + 2] * X[0][2]);
      ++nsol;
    }
  }
  return nsol;
}

/* ------------------------------------------------------------- scoring -- */
static void tree64(double* part, int stride, int nvals) {
  /* xor butterfly over 64 partials, for nvals interleaved quantities */
  double tmp[64];
  for (int v = 0; v < nvals; ++v) {
    for (int off = 32; off > 0; off >>= 1) {
      for (int l = 0; l < 64; ++l) tmp[l] = part[l * stride + v] + part[(l ^ off) * stride + v];
      for (int l = 0; l < 64; ++l) part[l * stride + v] = tmp[l];
    }
  }
}

/* squared reprojection error of point i under pose; returns 0 and sets *e2, or 1
 * if the point is behind the camera */
static int reproj(const double* pose, const double* K, const double* xy,
                  const double* xyz, double* e2, double* Xc, double* r) {
  const double* R = pose; const double* t = pose + 9;
  Xc[0] = R[0] * xyz[0] + R[1] * xyz[1] + R[2] * xyz[2] + t[0];
  Xc[1] = R[3] * xyz[0] + R[4] * xyz[1] + R[5] * xyz[2] + t[1];
  Xc[2] = R[6] * xyz[0] + R[7] * xyz[1] + R[8] * xyz[2] + t[2];
  if (!(Xc[2] > 0.0)) return 1;
  const double iz = 1.0 / Xc[2];
  const double px = (K[0] * Xc[0] + K[1] * Xc[1]) * iz + K[2];
  const double py = (K[4] * Xc[1]) * iz + K[5];
  r[0] = px - xy[0];
  r[1] = py - xy[1];
  *e2 = r[0] * r[0] + r[1] * r[1];
  return 0;
}

/* 256 partials: butterfly inside each group of 64, then (g0 + g1) + (g2 + g3) */
static void tree256(double* part, int stride, int nvals) {
  for (int g = 0; g < 4; ++g) tree64(part + (size_t)g * 64 * stride, stride, nvals);
  for (int v = 0; v < nvals; ++v)
    part[v] = (part[v] + part[64 * stride + v]) + (part[128 * stride + v] + part[192 * stride + v]);
}

/* 1024 partials (round 3: four workgroups of 256 per object in the local optimisation):
 * tree256 inside each group of 256, then (W0 + W1) + (W2 + W3) */
#define LO_P 1024
static void tree1024(double* part, int stride, int nvals) {
  for (int g = 0; g < 4; ++g) tree256(part + (size_t)g * 256 * stride, stride, nvals);
  for (int v = 0; v < nvals; ++v)
    part[v] = (part[v] + part[256 * stride + v]) + (part[512 * stride + v] + part[768 * stride + v]);
}

/* MSAC score and inlier count of a pose over the index list idx[0..m);
 * P = 64 (hypothesis scoring), 256 or 1024 (local optimisation) strided partials */
static double score_pose_p(const double* pose, const double* K, const double* xy,
                           const double* xyz, const int32_t* idx, int64_t m,
                           double thr2, int32_t* count, int P) {
  double part[LO_P];
  const double inv_thr2 = 1.0 / thr2;
  int32_t cnt = 0;
  for (int l = 0; l < P; ++l) {
    double acc = 0.0;
    for (int64_t i = l; i < m; i += P) {
      const int32_t p = idx[i];
      double e2, Xc[3], r[2];
      if (reproj(pose, K, xy + 2 * p, xyz + 3 * p, &e2, Xc, r)) continue;
      if (e2 < thr2) { acc += 1.0 - e2 * inv_thr2; ++cnt; }
    }
    part[l] = acc;
  }
  if (P == 64) tree64(part, 1, 1);
  else if (P == 256) tree256(part, 1, 1);
  else tree1024(part, 1, 1);
  *count = cnt;
  return part[0];
}
static double score_pose(const double* pose, const double* K, const double* xy,
                         const double* xyz, const int32_t* idx, int64_t m,
                         double thr2, int32_t* count) {
  return score_pose_p(pose, K, xy, xyz, idx, m, thr2, count, 64);
}
static double score_pose_lo(const double* pose, const double* K, const double* xy,
                            const double* xyz, const int32_t* idx, int64_t m,
                            double thr2, int32_t* count) {
  return score_pose_p(pose, K, xy, xyz, idx, m, thr2, count, LO_P);
}

/* b^e by binary exponentiation (multiplications only: the same bits everywhere) */
static double powi(double b, int64_t e) {
  double r = 1.0;
  while (e > 0) {
    if (e & 1) r = r * b;
    b = b * b;
    e >>= 1;
  }
  return r;
}

/* ------------------------------------------- spatial-coherence labelling -- */
#define GC_Q 1048576            /* 2^20 fixed point for the energies */
#define LO_MIN_GAIN 1e-6        /* relative MSAC gain below which the refits stop */

/* j is a neighbour of i iff both are active, i != j and their distance in
 * (x, y, s X, s Y, s Z) is at most tau_d (EPOS CVPR'20 sec. 3.4: tau_d = 20, s = 0.1). */
static int gc_neighbours(const double* xy, const double* xyz, int32_t a, int32_t b,
                         double s2, double r2) {
  const double dx = xy[2 * a] - xy[2 * b], dy = xy[2 * a + 1] - xy[2 * b + 1];
  const double dX = xyz[3 * a] - xyz[3 * b], dY = xyz[3 * a + 1] - xyz[3 * b + 1],
               dZ = xyz[3 * a + 2] - xyz[3 * b + 2];
  const double d2 = (dx * dx + dy * dy) + s2 * ((dX * dX + dY * dY) + dZ * dZ);
  return d2 <= r2;
}

/* ---- exact mode (gc_sweeps < 0): s-t minimum cut of the labelling energy ------------
 * The energy of gc_label below, times 2 Q GC_W / lambda so that every term is an integer:
 *   unary    inlier: 0 | w Q        outlier: w (Q - q_p) | 0     (q_p < Q | q_p == Q),
 *            w = round(GC_W 2 (1 - lambda) / lambda)   (18 GC_W at lambda = 0.1)
 *   pairwise (in, in) GC_W (2 Q - (q_p + q_q)),  (out, out) GC_W (q_p + q_q),  mixed 2 Q GC_W
 * It is submodular (V(0,0) + V(1,1) = 2 Q <= 4 Q = V(0,1) + V(1,0)), so the standard
 * construction (Kolmogorov & Zabih, PAMI 2004, table 1: theta(x_i, x_j) = A + (C - A) x_i +
 * (D - C) x_j + (B + C - A - D) [x_i = 0, x_j = 1]) turns it into a max-flow problem;
 * Dinic's algorithm on int64 capacities. Source side = outlier; of all minimum cuts the one
 * with the SMALLEST source side is returned (points not reachable from the source in the
 * residual graph are inliers) -- the convention of tests/test_oracle_fit.py's scipy twin.
 * The sweeps are what the HIP product runs; this mode exists so that the tests can say
 * where the two coincide (sparse graphs) and by how much they differ elsewhere. */
#define GC_W 1024
typedef struct { int32_t to, next; int64_t cap; } FlowEdge;
typedef struct { FlowEdge* e; int32_t* head; int32_t* level; int32_t* it; int32_t ne, nv; } FlowNet;

static void flow_add(FlowNet* g, int32_t a, int32_t b, int64_t cab, int64_t cba) {
  g->e[g->ne] = (FlowEdge){b, g->head[a], cab}; g->head[a] = g->ne++;
  g->e[g->ne] = (FlowEdge){a, g->head[b], cba}; g->head[b] = g->ne++;
}

static int flow_bfs(FlowNet* g, int32_t s, int32_t t, int32_t* queue) {
  for (int32_t v = 0; v < g->nv; ++v) g->level[v] = -1;
  int32_t qh = 0, qt = 0;
  queue[qt++] = s; g->level[s] = 0;
  while (qh < qt) {
    const int32_t v = queue[qh++];
    for (int32_t i = g->head[v]; i >= 0; i = g->e[i].next)
      if (g->e[i].cap > 0 && g->level[g->e[i].to] < 0) {
        g->level[g->e[i].to] = g->level[v] + 1;
        queue[qt++] = g->e[i].to;
      }
  }
  return g->level[t] >= 0;
}

/* one augmenting path of the level graph (iterative DFS: the graphs are thousands of nodes
 * deep in the worst case); returns the flow pushed, 0 when the level graph is exhausted */
static int64_t flow_dfs(FlowNet* g, int32_t s, int32_t t, int32_t* path) {
  int32_t depth = 0, v = s;
  for (;;) {
    if (v == t) {
      int64_t f = INT64_MAX;
      for (int32_t d = 0; d < depth; ++d) if (g->e[path[d]].cap < f) f = g->e[path[d]].cap;
      for (int32_t d = 0; d < depth; ++d) { g->e[path[d]].cap -= f; g->e[path[d] ^ 1].cap += f; }
      return f;
    }
    int advanced = 0;
    for (int32_t* i = &g->it[v]; *i >= 0; *i = g->e[*i].next) {
      const FlowEdge* ed = &g->e[*i];
      if (ed->cap > 0 && g->level[ed->to] == g->level[v] + 1) {
        path[depth++] = *i; v = ed->to; advanced = 1;
        break;
      }
    }
    if (advanced) continue;
    if (depth == 0) return 0;
    g->level[v] = -1;                                   /* dead end: prune */
    --depth;
    v = g->e[path[depth] ^ 1].to;
  }
}

static void gc_label_exact(const double* xy, const double* xyz, const int32_t* active,
                           int64_t n_active, double lam, double rad, double s2, double r2,
                           const int32_t* q, uint8_t* lab) {
  const int64_t m = n_active, Q = GC_Q;
  if (m < 1) return;
  const int64_t w = llround((double)GC_W * 2.0 * (1.0 - lam) / lam);
  /* neighbour pairs (i < j in active order) */
  int64_t n_pairs = 0, cap_pairs = 4 * m + 16;
  int32_t* pa = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)cap_pairs);
  for (int64_t i = 0; i < m; ++i)
    for (int64_t j = i + 1; j < m; ++j) {
      const int32_t p = active[i], o = active[j];
      if (fabs(xy[2 * p + 1] - xy[2 * o + 1]) > rad) continue;
      if (!gc_neighbours(xy, xyz, p, o, s2, r2)) continue;
      if (n_pairs == cap_pairs) {
        cap_pairs *= 2;
        pa = (int32_t*)realloc(pa, sizeof(int32_t) * 2 * (size_t)cap_pairs);
      }
      pa[2 * n_pairs] = (int32_t)i; pa[2 * n_pairs + 1] = (int32_t)j; ++n_pairs;
    }
  int64_t* th1 = (int64_t*)calloc((size_t)m, sizeof(int64_t));   /* cost of label inlier  */
  int64_t* th0 = (int64_t*)calloc((size_t)m, sizeof(int64_t));   /* cost of label outlier */
  for (int64_t i = 0; i < m; ++i) {
    const int64_t qp = q[active[i]];
    if (qp < Q) th0[i] = w * (Q - qp); else th1[i] = w * Q;
  }
  FlowNet g;
  g.nv = (int32_t)(m + 2);
  g.ne = 0;
  g.e = (FlowEdge*)malloc(sizeof(FlowEdge) * 2 * (size_t)(n_pairs + m));
  g.head = (int32_t*)malloc(sizeof(int32_t) * (size_t)g.nv);
  g.level = (int32_t*)malloc(sizeof(int32_t) * (size_t)g.nv);
  g.it = (int32_t*)malloc(sizeof(int32_t) * (size_t)g.nv);
  for (int32_t v = 0; v < g.nv; ++v) g.head[v] = -1;
  for (int64_t k = 0; k < n_pairs; ++k) {
    const int32_t i = pa[2 * k], j = pa[2 * k + 1];
    const int64_t sq = (int64_t)q[active[i]] + q[active[j]];
    const int64_t A = GC_W * sq, B = GC_W * 2 * Q, C = B, D = GC_W * (2 * Q - sq);
    th1[i] += C - A;
    th1[j] += D - C;
    flow_add(&g, i, j, B + C - A - D, 0);
  }
  const int32_t src = (int32_t)m, snk = (int32_t)m + 1;
  for (int64_t i = 0; i < m; ++i) {
    const int64_t diff = th1[i] - th0[i];
    if (diff > 0) flow_add(&g, src, (int32_t)i, diff, 0);
    else if (diff < 0) flow_add(&g, (int32_t)i, snk, -diff, 0);
  }
  int32_t* scratch = (int32_t*)malloc(sizeof(int32_t) * (size_t)g.nv);
  while (flow_bfs(&g, src, snk, scratch)) {
    for (int32_t v = 0; v < g.nv; ++v) g.it[v] = g.head[v];
    while (flow_dfs(&g, src, snk, scratch) > 0) {}
  }
  flow_bfs(&g, src, snk, scratch);                     /* level >= 0 <=> reachable from src */
  for (int64_t i = 0; i < m; ++i) lab[active[i]] = g.level[i] >= 0 ? 0 : 1;
  free(pa); free(th1); free(th0); free(g.e); free(g.head); free(g.level); free(g.it);
  free(scratch);
}

/* GC-RANSAC's labelling energy (Barath & Matas CVPR'18, eq. 2-3, in the form of the
 * published implementation's GCRANSAC::labeling): with d_p = min(e_p^2 / (1.5 tau_r)^2, 1)
 *   unary   U_p(inlier) = 0, U_p(outlier) = (1 - lambda)(1 - d_p)   if d_p < 1
 *           U_p(inlier) = (1 - lambda),  U_p(outlier) = 0            otherwise
 *   pairwise, neighbours p, q:  V(0,0) = lambda (d_p + d_q) / 2,  V(0,1) = V(1,0) = lambda,
 *                               V(1,1) = lambda (1 - (d_p + d_q) / 2)
 * A point is an inlier iff that is the cheaper label given its neighbours' labels:
 *   (1 - lambda) u_p + lambda T_p < 0,  u_p = -2 (Q - q_p) or +2 Q,
 *   T_p = 2 Q n0_p - (deg_p q_p + sum_{neighbours} q_j),  q = floor(d Q), Q = 2^20,
 * n0_p = number of neighbours currently labelled outlier. Synchronous sweeps from the
 * thresholded labelling [e^2 < tau_r^2]. lab[p]: 1 inlier, 0 outlier, 2 not active. */
static void gc_label(const double* pose, const double* K, const double* xy,
                     const double* xyz, const int32_t* active, int64_t n_active,
                     int64_t n, const PnpRefParams* prm, uint8_t* lab, uint8_t* tmp,
                     int32_t* q) {
  const double thr2 = prm->threshold * prm->threshold;
  const double tthr = 1.5 * prm->threshold, tthr2 = tthr * tthr;
  for (int64_t i = 0; i < n; ++i) lab[i] = 2;
  for (int64_t i = 0; i < n_active; ++i) {
    const int32_t p = active[i];
    double e2, Xc[3], r[2];
    if (reproj(pose, K, xy + 2 * p, xyz + 3 * p, &e2, Xc, r)) { q[p] = GC_Q; lab[p] = 0; continue; }
    double d = e2 / tthr2;
    if (!(d < 1.0)) d = 1.0;
    q[p] = (int32_t)(d * (double)GC_Q);
    lab[p] = e2 < thr2 ? 1 : 0;
  }
  const double lam = prm->spatial_coherence_weight;
  const double rad = prm->neighborhood_ball_radius;
  if (!(lam > 0.0) || !(rad > 0.0)) return;
  const double s2 = prm->scaling_from_millimeters * prm->scaling_from_millimeters;
  const double r2 = rad * rad;
  if (prm->gc_sweeps < 0) {       /* checker-only mode: the exact minimiser of the same energy */
    gc_label_exact(xy, xyz, active, n_active, lam, rad, s2, r2, q, lab);
    return;
  }
  for (int sweep = 0; sweep < prm->gc_sweeps; ++sweep) {
    for (int64_t i = 0; i < n_active; ++i) {
      const int32_t p = active[i];
      int64_t deg = 0, S = 0, n0 = 0;
      for (int64_t j = 0; j < n_active; ++j) {
        const int32_t o = active[j];
        if (o == p) continue;
        if (fabs(xy[2 * p + 1] - xy[2 * o + 1]) > rad) continue;     /* cannot be neighbours */
        if (!gc_neighbours(xy, xyz, p, o, s2, r2)) continue;
        ++deg; S += q[o]; n0 += lab[o] == 0;
      }
      const int64_t T = 2 * (int64_t)GC_Q * n0 - (deg * (int64_t)q[p] + S);
      const int64_t u = q[p] < GC_Q ? -2 * ((int64_t)GC_Q - q[p]) : 2 * (int64_t)GC_Q;
      const double val = (1.0 - lam) * (double)u + lam * (double)T;
      tmp[p] = val < 0.0 ? 1 : 0;
    }
    for (int64_t i = 0; i < n_active; ++i) lab[active[i]] = tmp[active[i]];
  }
}

/* ------------------------------------------------- local optimisation -- */
static void orthonormalize(double* R) {   /* Gram-Schmidt on rows, row2 = r0 x r1 */
  double n0 = sqrt(dot3(R, R));
  for (int j = 0; j < 3; ++j) R[j] /= n0;
  const double d = dot3(R, R + 3);
  for (int j = 0; j < 3; ++j) R[3 + j] -= d * R[j];
  const double n1 = sqrt(dot3(R + 3, R + 3));
  for (int j = 0; j < 3; ++j) R[3 + j] /= n1;
  cross3(R, R + 3, R + 6);
}

/* solve H x = -g for the symmetric positive (semi)definite 6x6 H = J^T J by Gaussian
 * elimination WITHOUT pivoting (stable for such matrices), one reciprocal per pivot
 * (used for its column and in the back-substitution); returns 0 on success, 1 when a pivot
 * is not positive. The HIP kernels run the same arithmetic (round 3: the pivot search of
 * the first version cost the GPU more than the elimination). */
static int solve6(double H[6][6], const double* g, double* x) {
  double A[6][7], inv[6];
  for (int i = 0; i < 6; ++i) { for (int j = 0; j < 6; ++j) A[i][j] = H[i][j]; A[i][6] = -g[i]; }
  int bad = 0;
  for (int c = 0; c < 6; ++c) {
    if (!(A[c][c] > 1e-300)) bad = 1;
    inv[c] = 1.0 / A[c][c];
    for (int r = c + 1; r < 6; ++r) {
      const double fct = A[r][c] * inv[c];
      for (int j = c + 1; j < 7; ++j) A[r][j] -= fct * A[c][j];
    }
  }
  if (bad) return 1;
  for (int i = 5; i >= 0; --i) {
    double s = A[i][6];
    for (int j = i + 1; j < 6; ++j) s -= A[i][j] * x[j];
    x[i] = s * inv[i];
  }
  return 0;
}

/* one Gauss-Newton step on the inliers of `pose` -- at thr2 (lab == NULL), or the
 * points lab[p] == want (the labelled set: fixed membership, full weight) -- writes
 * `next` */
static int gn_step_sel(const double* pose, const double* K, const double* xy,
                       const double* xyz, const int32_t* idx, int64_t m, double thr2,
                       const uint8_t* lab, int want, double* next, int P) {
  /* 27 accumulated quantities: 21 upper-triangular H entries + 6 of g; P = 1024 strided
   * partials in the local optimisation of a proposal, 256 in the joint refinement */
  static double part[LO_P * 27];
  for (int l = 0; l < P; ++l) {
    double acc[27];
    for (int v = 0; v < 27; ++v) acc[v] = 0.0;
    for (int64_t i = l; i < m; i += P) {
      const int32_t p = idx[i];
      double e2, Xc[3], r[2];
      if (reproj(pose, K, xy + 2 * p, xyz + 3 * p, &e2, Xc, r)) continue;
      if (lab ? lab[p] != want : !(e2 < thr2)) continue;
      const double iz = 1.0 / Xc[2];
      /* d(px,py)/dXc */
      const double a0 = K[0] * iz, a1 = K[1] * iz,
                   a2 = -(K[0] * Xc[0] + K[1] * Xc[1]) * iz * iz;
      const double b1 = K[4] * iz, b2 = -(K[4] * Xc[1]) * iz * iz;
      /* dXc/dw = -[Xc]x ; J = Jpi * [-[Xc]x | I] */
      double J0[6], J1[6];
      /* Xc(w) = (I + [w]x) Xc = Xc - [Xc]x w, with
       * -[Xc]x = [[0, z, -y], [-z, 0, x], [y, -x, 0]]: row . (-[Xc]x).
       * (Round 1 had these six entries with the opposite sign -- the steps were then
       * never accepted and the local optimisation was a no-op; found in round 2 by
       * tests/test_oracle_fit.py's finite-difference stationarity check.) */
      J0[0] = -a1 * Xc[2] + a2 * Xc[1];
      J0[1] = a0 * Xc[2] - a2 * Xc[0];
      J0[2] = -a0 * Xc[1] + a1 * Xc[0];
      J0[3] = a0; J0[4] = a1; J0[5] = a2;
      J1[0] = -b1 * Xc[2] + b2 * Xc[1];
      J1[1] = -b2 * Xc[0];
      J1[2] = b1 * Xc[0];
      J1[3] = 0.0; J1[4] = b1; J1[5] = b2;
      int v = 0;
      for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) acc[v++] += J0[a] * J0[b] + J1[a] * J1[b];
      for (int a = 0; a < 6; ++a) acc[v++] += J0[a] * r[0] + J1[a] * r[1];
    }
    for (int v = 0; v < 27; ++v) part[l * 27 + v] = acc[v];
  }
  if (P == 256) tree256(part, 27, 27); else tree1024(part, 27, 27);
  double H[6][6], g[6], x[6];
  int v = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) { H[a][b] = part[v]; H[b][a] = part[v]; ++v; }
  for (int a = 0; a < 6; ++a) g[a] = part[v++];
  if (solve6(H, g, x)) return 1;
  /* dR from the unit quaternion (1, w/2) / |.| */
  double qw = 1.0, qx = 0.5 * x[0], qy = 0.5 * x[1], qz = 0.5 * x[2];
  const double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  qw /= qn; qx /= qn; qy /= qn; qz /= qn;
  double dR[9];
  dR[0] = 1.0 - 2.0 * (qy * qy + qz * qz); dR[1] = 2.0 * (qx * qy - qz * qw); dR[2] = 2.0 * (qx * qz + qy * qw);
  dR[3] = 2.0 * (qx * qy + qz * qw); dR[4] = 1.0 - 2.0 * (qx * qx + qz * qz); dR[5] = 2.0 * (qy * qz - qx * qw);
  dR[6] = 2.0 * (qx * qz - qy * qw); dR[7] = 2.0 * (qy * qz + qx * qw); dR[8] = 1.0 - 2.0 * (qx * qx + qy * qy);
  const double* R = pose; const double* t = pose + 9;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      next[i * 3 + j] = dR[i * 3] * R[j] + dR[i * 3 + 1] * R[3 + j] + dR[i * 3 + 2] * R[6 + j];
    next[9 + i] = dR[i * 3] * t[0] + dR[i * 3 + 1] * t[1] + dR[i * 3 + 2] * t[2] + x[3 + i];
  }
  for (int i = 0; i < 12; ++i) if (!(next[i] == next[i])) return 1;   /* NaN */
  return 0;
}

static int gn_step(const double* pose, const double* K, const double* xy,
                   const double* xyz, const int32_t* idx, int64_t m, double thr2,
                   double* next) {
  return gn_step_sel(pose, K, xy, xyz, idx, m, thr2, NULL, 0, next, LO_P);
}

/* ------------------------------------------------ joint refinement (PEARL) -- */
/* The role PEARL plays in Progressive-X (Isack & Boykov IJCV'12; Barath & Matas
 * ICCV'19 sec. 3.3), for k instances, over ALL n correspondences:
 *   labels l_p in {0..k-1, k = outlier};  data term D_p(m) = min(e_pm^2 / (1.5 tau)^2, 1)
 *   for an instance, D_p(outlier) = (tau / (1.5 tau))^2 (a point is worth explaining iff
 *   its error is below tau); Potts smoothness on the neighbourhood graph, NORMALISED by
 *   the degree: a point pays lambda times the FRACTION of its neighbours that carry
 *   another label (EPOS's many-to-many correspondences give every inlier dozens of
 *   outlier neighbours -- the symmetric / wrong candidates of the same pixels -- and an
 *   un-normalised Potts term then relabels everything as outlier).
 *   Energy E = sum_p [(1 - lambda) D_p(l_p) + lambda floor(Q diff_p / deg_p)] in 2^-20
 *   fixed point (exact integers: order independent).
 * One iteration = gc_sweeps synchronous relabelling sweeps (each point takes its
 * cheapest label given its neighbours' labels; ties -> lowest label) + one Gauss-Newton
 * refit of every instance on its points. The new poses are kept iff E dropped. */
static int64_t pearl_energy(const int64_t* D /*[n][k+1]*/, const uint8_t* lab, int64_t n,
                            int k, const double* xy, const double* xyz, double lam,
                            double s2, double r2, double rad, int64_t* smooth_out) {
  int64_t data = 0, smooth = 0;
  for (int64_t p = 0; p < n; ++p) {
    data += D[p * (k + 1) + lab[p]];
    int64_t deg = 0, diff = 0;
    for (int64_t o = 0; o < n; ++o) {
      if (o == p || fabs(xy[2 * p + 1] - xy[2 * o + 1]) > rad) continue;
      if (!gc_neighbours(xy, xyz, (int32_t)p, (int32_t)o, s2, r2)) continue;
      ++deg; diff += lab[o] != lab[p];
    }
    if (deg > 0) smooth += ((int64_t)GC_Q * diff) / deg;   /* fraction of disagreeing neighbours */
  }
  (void)lam;
  *smooth_out = smooth;
  return data;
}

static void pearl_refine(double* poses, int k, const double* K, const double* xy,
                         const double* xyz, int64_t n, const PnpRefParams* prm,
                         int32_t* labels) {
  const double lam = prm->spatial_coherence_weight, rad = prm->neighborhood_ball_radius;
  if (k < 2 || k > 8 || k > prm->max_model_number_for_optimization || !(lam > 0.0) ||
      !(rad > 0.0) || prm->gc_sweeps == 0)
    return;
  const double tthr = 1.5 * prm->threshold, tthr2 = tthr * tthr;
  const double thr2 = prm->threshold * prm->threshold;
  const double s2 = prm->scaling_from_millimeters * prm->scaling_from_millimeters;
  const double r2 = rad * rad;
  int64_t* D = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n * (k + 1)));
  uint8_t* lab = (uint8_t*)malloc((size_t)n);
  uint8_t* tmp = (uint8_t*)malloc((size_t)n);
  int32_t* all = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  double cand[8 * 12];
  for (int64_t i = 0; i < n; ++i) all[i] = (int32_t)i;
  const int64_t d_out = (int64_t)((thr2 / tthr2) * (double)GC_Q);
  for (int it = 0; it < prm->pearl_iters; ++it) {
    /* data terms under the current poses; start from the current labels */
    for (int64_t p = 0; p < n; ++p) {
      for (int m = 0; m < k; ++m) {
        double e2, Xc[3], r[2];
        int64_t dq = GC_Q;
        if (!reproj(poses + 12 * m, K, xy + 2 * p, xyz + 3 * p, &e2, Xc, r)) {
          double d = e2 / tthr2;
          if (!(d < 1.0)) d = 1.0;
          dq = (int64_t)(d * (double)GC_Q);
        }
        D[p * (k + 1) + m] = dq;
      }
      D[p * (k + 1) + k] = d_out;
      lab[p] = labels[p] >= 0 && labels[p] < k ? (uint8_t)labels[p] : (uint8_t)k;
    }
    int64_t sm0;
    const int64_t da0 = pearl_energy(D, lab, n, k, xy, xyz, lam, s2, r2, rad, &sm0);
    const double e_before = (1.0 - lam) * (double)da0 + lam * (double)sm0;
    for (int sweep = 0; sweep < (prm->gc_sweeps < 0 ? 2 : prm->gc_sweeps); ++sweep) {
      for (int64_t p = 0; p < n; ++p) {
        int64_t cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int64_t deg = 0;
        for (int64_t o = 0; o < n; ++o) {
          if (o == p || fabs(xy[2 * p + 1] - xy[2 * o + 1]) > rad) continue;
          if (!gc_neighbours(xy, xyz, (int32_t)p, (int32_t)o, s2, r2)) continue;
          ++deg; ++cnt[lab[o]];
        }
        int best = 0;
        double best_c = 0.0;
        for (int m = 0; m <= k; ++m) {
          const double c = (1.0 - lam) * (double)D[p * (k + 1) + m] +
                           lam * (double)(deg > 0 ? ((int64_t)GC_Q * (deg - cnt[m])) / deg : 0);
          if (m == 0 || c < best_c) { best = m; best_c = c; }
        }
        tmp[p] = (uint8_t)best;
      }
      memcpy(lab, tmp, (size_t)n);
    }
    /* refit every instance on its points (one Gauss-Newton step, full weight) */
    int moved = 0;
    for (int m = 0; m < k; ++m) {
      memcpy(cand + 12 * m, poses + 12 * m, sizeof(double) * 12);
      int64_t cntm = 0;
      for (int64_t p = 0; p < n; ++p) cntm += lab[p] == m;
      if (cntm < prm->min_point_number) continue;
      double next[12];
      if (!gn_step_sel(poses + 12 * m, K, xy, xyz, all, n, thr2, lab, m, next, 256)) {
        memcpy(cand + 12 * m, next, sizeof(next));
        moved = 1;
      }
    }
    if (getenv("PNP_REF_DEBUG")) {
      int64_t c[10] = {0};
      for (int64_t p = 0; p < n; ++p) ++c[lab[p]];
      fprintf(stderr, "pearl it %d: moved %d label counts %lld %lld %lld, before %.6g (data %lld smooth %lld)\n",
              it, moved, (long long)c[0], (long long)c[1], (long long)c[2], e_before, (long long)da0, (long long)sm0);
    }
    if (!moved) break;
    /* energy of (new poses, new labels) against (old poses, old labels) */
    for (int64_t p = 0; p < n; ++p)
      for (int m = 0; m < k; ++m) {
        double e2, Xc[3], r[2];
        int64_t dq = GC_Q;
        if (!reproj(cand + 12 * m, K, xy + 2 * p, xyz + 3 * p, &e2, Xc, r)) {
          double d = e2 / tthr2;
          if (!(d < 1.0)) d = 1.0;
          dq = (int64_t)(d * (double)GC_Q);
        }
        D[p * (k + 1) + m] = dq;
      }
    int64_t sm1;
    const int64_t da1 = pearl_energy(D, lab, n, k, xy, xyz, lam, s2, r2, rad, &sm1);
    const double e_after = (1.0 - lam) * (double)da1 + lam * (double)sm1;
    if (getenv("PNP_REF_DEBUG")) fprintf(stderr, "pearl it %d: before %.6g (data %lld smooth %lld) after %.6g (data %lld smooth %lld)\n", it, e_before, (long long)da0, (long long)sm0, e_after, (long long)da1, (long long)sm1);
    if (!(e_after < e_before)) break;
    memcpy(poses, cand, sizeof(double) * 12 * (size_t)k);
    for (int64_t p = 0; p < n; ++p) labels[p] = lab[p] < k ? (int32_t)lab[p] : -1;
  }
  free(D); free(lab); free(tmp); free(all);
}

/* ----------------------------------------------------------- main entry -- */
static void bearing(const double* K, const double* xy, double* f) {
  /* K^-1 [x y 1] for upper-triangular K, normalised */
  const double y = (xy[1] - K[5]) / K[4];
  const double x = (xy[0] - K[2] - K[1] * y) / K[0];
  const double n = sqrt(x * x + y * y + 1.0);
  f[0] = x / n; f[1] = y / n; f[2] = 1.0 / n;
}

/* Returns the number of poses found (k), fills poses[k*12], labels[n], scores[k]. */
int pnp_ref_find6d_poses(const double* xy, const double* xyz, int64_t n,
                         const double* K, const PnpRefParams* prm, uint64_t seed,
                         double* poses, int32_t* labels, double* scores,
                         int32_t max_k) {
  for (int64_t i = 0; i < n; ++i) labels[i] = -1;
  if (n < prm->min_point_number || n < 3) return 0;      /* infer.py:420-422 */
  int want = prm->max_model_number;
  if (want < 0 || want > max_k) want = max_k;
  const double thr2 = prm->threshold * prm->threshold;
  const int64_t words = (n + 63) / 64;
  int32_t* active = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* all = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  uint64_t* inl = (uint64_t*)calloc((size_t)(words * (max_k + 1)), sizeof(uint64_t));
  int64_t n_active = n;
  for (int64_t i = 0; i < n; ++i) { active[i] = (int32_t)i; all[i] = (int32_t)i; }
  uint8_t* lab = (uint8_t*)malloc((size_t)n);
  uint8_t* lab_tmp = (uint8_t*)malloc((size_t)n);
  int32_t* gq = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int k = 0;
  /* Progressive-X mode (more than one instance wanted): a failed proposal is retried
   * with fresh samples under the `conf` criterion, within two extra rounds */
  const int max_rounds = want + (want > 1 ? 2 : 0);
  int tries = 0;
  int64_t last_new = prm->min_point_number;
  for (int round = 0; round < max_rounds && k < want; ++round) {
    if (n_active < prm->min_point_number || n_active < 3) break;
    double best_pose[12];
    double best_score = -1.0;
    int32_t best_count = 0;
    int failed = 0;
    for (int it = 0; it < prm->max_iters; ++it) {
      /* RANSAC termination at confidence proposal_engine_conf (1.0: never) */
      if (it > 0 && prm->proposal_engine_conf < 1.0 && best_score > 0.0) {
        const double w = (double)best_count / (double)n_active;
        if (powi(1.0 - w * w * w, it) <= 1.0 - prm->proposal_engine_conf) break;
      }
      int64_t m = n_active;
      if (prm->use_prosac) {   /* growing prefix of the (confidence-sorted) list */
        m = (n_active * (int64_t)(it + 1) + prm->max_iters - 1) / prm->max_iters;
        if (m < prm->min_point_number) m = prm->min_point_number;
        if (m < 3) m = 3;
        if (m > n_active) m = n_active;
      }
      int64_t s[3];
      sample3(seed, (uint32_t)round, (uint32_t)it, m, s);
      double f[3][3], X[3][3], p2[3][2];
      for (int j = 0; j < 3; ++j) {
        const int32_t p = active[s[j]];
        bearing(K, xy + 2 * p, f[j]);
        for (int d = 0; d < 3; ++d) X[j][d] = xyz[3 * p + d];
        p2[j][0] = xy[2 * p]; p2[j][1] = xy[2 * p + 1];
      }
      const double area = 0.5 * fabs((p2[1][0] - p2[0][0]) * (p2[2][1] - p2[0][1]) -
                                     (p2[1][1] - p2[0][1]) * (p2[2][0] - p2[0][0]));
      if (area < prm->min_triangle_area) continue;
      double sols[48];
      const int ns = p3p(f, X, sols);
      for (int q = 0; q < ns; ++q) {
        int32_t cnt;
        const double sc = score_pose(sols + 12 * q, K, xy, xyz, active, n_active, thr2, &cnt);
        if (sc > best_score) {
          best_score = sc; best_count = cnt;
          memcpy(best_pose, sols + 12 * q, sizeof(best_pose));
        }
      }
    }
    if (!(best_score > 0.0) || best_count < 3) failed = 1;
    if (!failed) {
      /* local optimisation (i): refits on the inliers at the threshold */
      orthonormalize(best_pose);
      {
        int32_t cnt;
        best_score = score_pose_lo(best_pose, K, xy, xyz, active, n_active, thr2, &cnt);
        best_count = cnt;
      }
      for (int li = 0; li < prm->lo_iters; ++li) {
        double cand[12];
        if (gn_step(best_pose, K, xy, xyz, active, n_active, thr2, cand)) break;
        int32_t cnt;
        const double sc = score_pose_lo(cand, K, xy, xyz, active, n_active, thr2, &cnt);
        if (!(sc > best_score)) break;
        const double gain = sc - best_score;
        best_score = sc; best_count = cnt;
        memcpy(best_pose, cand, sizeof(cand));
        if (!(gain > LO_MIN_GAIN * sc)) break;      /* converged: further steps are noise */
      }
      /* (ii): spatially coherent inlier set -> refits on it */
      if (prm->gc_sweeps != 0 && prm->spatial_coherence_weight > 0.0 &&
          prm->neighborhood_ball_radius > 0.0) {
        gc_label(best_pose, K, xy, xyz, active, n_active, n, prm, lab, lab_tmp, gq);
        for (int li = 0; li < prm->lo_iters; ++li) {
          double cand[12];
          if (gn_step_sel(best_pose, K, xy, xyz, active, n_active, thr2, lab, 1, cand, LO_P)) break;
          int32_t cnt;
          const double sc = score_pose_lo(cand, K, xy, xyz, active, n_active, thr2, &cnt);
          if (!(sc > best_score)) break;
          const double gain = sc - best_score;
          best_score = sc; best_count = cnt;
          memcpy(best_pose, cand, sizeof(cand));
          if (!(gain > LO_MIN_GAIN * sc)) break;
        }
      }
      if (best_count < prm->min_point_number) failed = 1;
    }
    /* inliers over ALL correspondences, Tanimoto / coverage tests */
    uint64_t* cur = inl + (size_t)k * words;
    memset(cur, 0, sizeof(uint64_t) * (size_t)words);
    int64_t n_inl = 0, n_new = 0;
    for (int64_t i = 0; i < n && !failed; ++i) {
      double e2, Xc[3], r[2];
      if (reproj(best_pose, K, xy + 2 * i, xyz + 3 * i, &e2, Xc, r)) continue;
      if (e2 < thr2) {
        cur[i >> 6] |= 1ULL << (i & 63);
        ++n_inl;
        if (labels[i] < 0) ++n_new;
      }
    }
    int ok = !failed && n_inl > 0;
    for (int j = 0; j < k && ok; ++j) {
      const uint64_t* pj = inl + (size_t)j * words;
      int64_t inter = 0, uni = 0;
      for (int64_t w = 0; w < words; ++w) {
        inter += __builtin_popcountll(cur[w] & pj[w]);
        uni += __builtin_popcountll(cur[w] | pj[w]);
      }
      if ((double)inter >= prm->max_tanimoto_similarity * (double)uni) ok = 0;
    }
    if (ok && (double)n_new < prm->min_coverage * (double)n_inl) ok = 0;
    if (!ok) {
      /* Progressive-X termination: give up once the samples drawn since the last success
       * would have hit an instance as large as the last accepted one (or, before any,
       * a minimal one) with probability >= conf */
      ++tries;
      if (want == 1) break;
      const int64_t ref = k == 0 ? prm->min_point_number : last_new;
      if (ref >= n_active) break;
      const double w = (double)ref / (double)n_active;
      if (!(powi(1.0 - w * w * w, (int64_t)tries * prm->max_iters) > 1.0 - prm->conf)) break;
      continue;
    }
    tries = 0;
    last_new = n_new;
    memcpy(poses + 12 * k, best_pose, sizeof(best_pose));
    scores[k] = best_score;
    /* label and remove the inliers that were still active (stable compaction) */
    int64_t w = 0;
    for (int64_t i = 0; i < n_active; ++i) {
      const int32_t p = active[i];
      if ((cur[p >> 6] >> (p & 63)) & 1ULL) labels[p] = k;
      else active[w++] = p;
    }
    n_active = w;
    ++k;
  }
  if (prm->pearl_iters > 0) pearl_refine(poses, k, K, xy, xyz, n, prm, labels);
  free(active); free(all); free(inl); free(lab); free(lab_tmp); free(gq);
  return k;
}

void pnp_ref_params_default(PnpRefParams* p) {   /* scripts/infer.py:76-120,470-488 */
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

/* exposed for unit tests: the spatial-coherence labelling of `pose` over all n
 * correspondences (all active); lab_out[n] in {0, 1} */
void pnp_ref_gc_label(const double* pose, const double* K, const double* xy,
                      const double* xyz, int64_t n, const PnpRefParams* prm,
                      uint8_t* lab_out) {
  int32_t* active = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
  uint8_t* tmp = (uint8_t*)malloc((size_t)(n + 1));
  int32_t* q = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
  for (int64_t i = 0; i < n; ++i) active[i] = (int32_t)i;
  gc_label(pose, K, xy, xyz, active, n, n, prm, lab_out, tmp, q);
  free(active); free(tmp); free(q);
}
double pnp_ref_powi(double b, int64_t e) { return powi(b, e); }

/* exposed for unit tests */
int pnp_ref_p3p(const double* f9, const double* X9, double* pose48) {
  double f[3][3], X[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { f[i][j] = f9[i * 3 + j]; X[i][j] = X9[i * 3 + j]; }
  return p3p(f, X, pose48);
}
double pnp_ref_cubic_root(double b, double c, double d) { return cubic_root(b, c, d); }
