/*
 * ORACLE (test infrastructure, not product): plain-C, single-thread restatement of the
 * reference's alternative fitting method
 *     cv2.solvePnPRansac(objectPoints, imagePoints, K, None, iterationsCount=400,
 *                        reprojectionError=4.0, confidence=0.99, flags=cv2.SOLVEPNP_EPNP)
 *                                                          scripts/infer.py:505-528
 * (one instance per object, score 0.0, R = Rodrigues(rvec)).
 *
 * PARITY UNPINNED. The arithmetic lives in OpenCV (README.md:29 pins `opencv=3.4.2`, a
 * conda package that is not vendored and not installed here: `import cv2` fails), and the
 * reference holds no test or golden pose for this call. This file restates the PUBLISHED
 * behaviour of that version's calib3d module (solvePnPRansac / RANSACPointSetRegistrator /
 * the EPnP solver of Lepetit, Moreno-Noguer & Fua, IJCV 2009) from its documented algorithm:
 *
 *   solvePnPRansac: 2D and 3D points are rounded to float32 on entry. Minimal sets of 5
 *     correspondences (model_points = 5 for EPNP), drawn by cv::RNG (multiply-with-carry,
 *     state 2^64 - 1, `next() % n`, re-drawing an index that is already in the set); EPnP on
 *     the 5 points gives ONE model per set; its inliers are the points whose squared
 *     reprojection error, evaluated in float32 from the float32 projection, is <= (float)
 *     (reprojectionError^2) (no cheirality test); the model with strictly more inliers than
 *     the best so far (and more than 4) becomes the best and the iteration cap shrinks to
 *     round(log(1 - conf) / log(1 - w^5)), w = its inlier ratio, w^5 rounded once as
 *     std::pow returns it; afterwards EPnP is run once more on ALL inliers of the best model
 *     and that pose is returned. The SOLVER (not the inlier test) sees every image point
 *     after solvePnP's undistortPoints / init_points round trip: (float)((u - cx) / fx)
 *     mapped back by x * fu + uc (round 4; before, the solver saw the float32 pixels).
 *   EPnP: 4 control points (centroid + principal directions scaled by sqrt(eigenvalue / n);
 *     here by decreasing eigenvalue and signed so that the largest component is positive),
 *     barycentric coordinates, the 12 x 12 matrix M^T M, its four eigenvectors of smallest
 *     eigenvalue, the 6 x 10 system of control-point distances, the three closed-form beta
 *     initialisations (N = 1, 2, 3 of the paper), 5 Gauss-Newton iterations each, absolute
 *     orientation of the camera-frame points (R = U V^T of sum (pc - pc0)(pw - pw0)^T, third
 *     row negated if det < 0), the candidate of smallest mean reprojection error wins.
 *
 * What differs from OpenCV on purpose, so that a wavefront-parallel twin can reproduce the
 * bits: the dense factorisations are the build's own (Jacobi for the symmetric
 * eigenproblems -- cyclic for 3 x 3, round-robin order for 12 x 12 --, Householder QR for the 6 x k least-squares systems, one-sided Jacobi for the
 * 3 x 3 SVD) instead of cv::SVD / LAPACK; sums over correspondences that are linear in the
 * points (camera-frame centroid, the 3 x 3 cross-covariance) are formed from the centroid and
 * covariance of the object points instead of per point; the rvec <-> R round trip through
 * Rodrigues is skipped (cv::Rodrigues orthonormalises by cv::SVD and goes through acos /
 * sin / cos of the C library: not restatable bit for bit, and a device libm and a host libm
 * need not agree -- so the inlier test here sees EPnP's R itself where OpenCV's sees
 * Rodrigues(Rodrigues(R)), equal to ~1e-16; scripts that use this method are told so at
 * run time); sums over the inliers of the final fit use the canonical order of
 * pnp_ref.c (256 strided partials, butterfly, (g0 + g1) + (g2 + g3)). Only + - * / sqrt are
 * used, except log() in the iteration bound; build with -ffp-contract=off.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EP_JAC_SWEEPS 12
#define EP_MODEL_POINTS 5

/* ------------------------------------------------------------------ cv::RNG -- */
static uint32_t cvrng_next(uint64_t* st) {
  *st = (uint64_t)(uint32_t)(*st) * 4164903690u + (uint32_t)(*st >> 32);
  return (uint32_t)(*st);
}

/* ------------------------------------------------------- small dense algebra -- */
/* One Jacobi rotation (p, q) of the symmetric n x n matrix A (row-major, both triangles
 * kept), accumulated into V. Returns 1 if a rotation was applied. */
static int jacobi_rotate(int n, double* A, double* V, int p, int q) {
  const double apq = A[p * n + q];
  if (apq == 0.0) return 0;
  const double app = A[p * n + p], aqq = A[q * n + q];
  if (fabs(apq) <= 8.673617379884035e-19 * (fabs(app) + fabs(aqq))) {   /* 2^-60 */
    A[p * n + q] = 0.0; A[q * n + p] = 0.0;
    return 0;
  }
  const double theta = (aqq - app) / (2.0 * apq);
  const double at = fabs(theta);
  double t = 1.0 / (at + sqrt(theta * theta + 1.0));
  if (theta < 0.0) t = -t;
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
  for (int k = 0; k < n; ++k) {
    if (k != p && k != q) {
      const double akp = A[k * n + p], akq = A[k * n + q];
      const double np_ = c * akp - s * akq, nq_ = s * akp + c * akq;
      A[k * n + p] = np_; A[p * n + k] = np_;
      A[k * n + q] = nq_; A[q * n + k] = nq_;
    }
    const double vkp = V[k * n + p], vkq = V[k * n + q];
    V[k * n + p] = c * vkp - s * vkq;
    V[k * n + q] = s * vkp + c * vkq;
  }
  A[p * n + p] = app - t * apq;
  A[q * n + q] = aqq + t * apq;
  A[p * n + q] = 0.0; A[q * n + p] = 0.0;
  return 1;
}

/* Cyclic Jacobi (row by row). On return the diagonal of A holds the eigenvalues and column
 * j of V the eigenvector of A[j][j]. */
static void jacobi_sym(int n, double* A, double* V) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < EP_JAC_SWEEPS; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) rotated |= jacobi_rotate(n, A, V, p, q);
    if (!rotated) break;
  }
}

/* The 12 x 12 problem of EPnP: the same rotations in ROUND-ROBIN order -- a sweep is 11
 * rounds of 6 disjoint pairs (round r: (r, 11) and ((r + m) mod 11, (r - m) mod 11),
 * m = 1..5), so that the GPU twin can apply the six rotations of a round at once (disjoint
 * rotations commute; applying them one after the other in the order m = 0..5, as here,
 * fixes the rounding of the entries two of them touch). */
static void jacobi12_rr(double* A, double* V) {
  for (int i = 0; i < 144; ++i) V[i] = (i % 13 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < EP_JAC_SWEEPS; ++sweep) {
    int rotated = 0;
    for (int r = 0; r < 11; ++r)
      for (int m = 0; m < 6; ++m) {
        /* no sorting: a rotation gives the same bits with the roles of p and q swapped */
        const int p = m == 0 ? r : (r + m) % 11, q = m == 0 ? 11 : (r - m + 11) % 11;
        rotated |= jacobi_rotate(12, A, V, p, q);
      }
    if (!rotated) break;
  }
}

/* min ||A x - b|| for a 6 x k system (k <= 5) by Householder QR. A (row-major, leading
 * dimension 5) and b are destroyed. Returns 1 when a column is exactly dependent. */
static int qr_solve6(int k, double A[6][5], double* b, double* x) {
  for (int j = 0; j < k; ++j) {
    double nrm2 = 0.0;
    for (int i = j; i < 6; ++i) nrm2 += A[i][j] * A[i][j];
    if (!(nrm2 > 0.0)) return 1;
    const double alpha = A[j][j] > 0.0 ? -sqrt(nrm2) : sqrt(nrm2);
    double v[6];
    for (int i = j; i < 6; ++i) v[i] = A[i][j];
    v[j] = v[j] - alpha;
    double vn2 = 0.0;
    for (int i = j; i < 6; ++i) vn2 += v[i] * v[i];
    if (!(vn2 > 0.0)) return 1;
    for (int c = j + 1; c < k; ++c) {
      double d = 0.0;
      for (int i = j; i < 6; ++i) d += v[i] * A[i][c];
      const double f = 2.0 * d / vn2;
      for (int i = j; i < 6; ++i) A[i][c] = A[i][c] - f * v[i];
    }
    double d = 0.0;
    for (int i = j; i < 6; ++i) d += v[i] * b[i];
    const double f = 2.0 * d / vn2;
    for (int i = j; i < 6; ++i) b[i] = b[i] - f * v[i];
    A[j][j] = alpha;
  }
  for (int j = k - 1; j >= 0; --j) {
    double s = b[j];
    for (int c = j + 1; c < k; ++c) s -= A[j][c] * x[c];
    x[j] = s / A[j][j];
  }
  return 0;
}

/* M = U diag(S) V^T for a 3 x 3 matrix by one-sided Jacobi; a vanishing singular direction
 * of U is completed with the cross product of the other two. */
static void svd3(const double* M, double* U, double* V) {
  double G[9];
  for (int i = 0; i < 9; ++i) { G[i] = M[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
  for (int sweep = 0; sweep < EP_JAC_SWEEPS; ++sweep) {
    int rotated = 0;
    for (int r = 0; r < 3; ++r) {
      const int p = PQ[r][0], q = PQ[r][1];
      const double al = G[p] * G[p] + G[3 + p] * G[3 + p] + G[6 + p] * G[6 + p];
      const double be = G[q] * G[q] + G[3 + q] * G[3 + q] + G[6 + q] * G[6 + q];
      const double ga = G[p] * G[q] + G[3 + p] * G[3 + q] + G[6 + p] * G[6 + q];
      if (ga == 0.0 || ga * ga <= 1e-34 * (al * be)) continue;
      const double zeta = (be - al) / (2.0 * ga);
      double t = 1.0 / (fabs(zeta) + sqrt(zeta * zeta + 1.0));
      if (zeta < 0.0) t = -t;
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; ++k) {
        const double gp = G[3 * k + p], gq = G[3 * k + q];
        G[3 * k + p] = c * gp - s * gq; G[3 * k + q] = s * gp + c * gq;
        const double vp = V[3 * k + p], vq = V[3 * k + q];
        V[3 * k + p] = c * vp - s * vq; V[3 * k + q] = s * vp + c * vq;
      }
      rotated = 1;
    }
    if (!rotated) break;
  }
  double S[3];
  for (int j = 0; j < 3; ++j) S[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
  int jmin = 0;
  if (S[1] < S[jmin]) jmin = 1;
  if (S[2] < S[jmin]) jmin = 2;
  double smax = S[0] > S[1] ? S[0] : S[1];
  if (S[2] > smax) smax = S[2];
  for (int j = 0; j < 3; ++j) {
    const double inv = 1.0 / S[j];
    U[j] = G[j] * inv; U[3 + j] = G[3 + j] * inv; U[6 + j] = G[6 + j] * inv;
  }
  if (!(S[jmin] > 1e-12 * smax)) {
    const int a = (jmin + 1) % 3, b = (jmin + 2) % 3;
    U[jmin] = U[3 + a] * U[6 + b] - U[6 + a] * U[3 + b];
    U[3 + jmin] = U[6 + a] * U[b] - U[a] * U[6 + b];
    U[6 + jmin] = U[a] * U[3 + b] - U[3 + a] * U[b];
  }
}

static int inv3(const double* m, double* inv) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8],
               c02 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  if (!(fabs(det) > 0.0)) return 1;
  const double id = 1.0 / det;
  inv[0] = c00 * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  inv[3] = c01 * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  inv[6] = c02 * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return 0;
}

/* ------------------------------------------------------- canonical summation -- */
static void tree64(double* part, int stride, int nvals) {
  for (int off = 32; off > 0; off >>= 1)
    for (int l = 0; l < 64; ++l)
      if ((l & off) == 0)
        for (int v = 0; v < nvals; ++v) {
          const double a = part[l * stride + v], b = part[(l | off) * stride + v];
          part[l * stride + v] = a + b; part[(l | off) * stride + v] = b + a;
        }
}
static void tree256(double* part, int stride, int nvals) {
  for (int g = 0; g < 4; ++g) tree64(part + (size_t)g * 64 * stride, stride, nvals);
  for (int v = 0; v < nvals; ++v)
    part[v] = (part[v] + part[64 * stride + v]) + (part[128 * stride + v] + part[192 * stride + v]);
}

/* sum_i f(idx[i]) for nvals values at once; P = 1 (the minimal sets: plain left-to-right)
 * or 256 (the final fit: strided partials + tree) */
typedef void (*TermFn)(const void* ctx, int32_t p, double* out);
static void canon_sum(TermFn fn, const void* ctx, const int32_t* idx, int64_t m, int P,
                      int nvals, double* out) {
  double term[48];
  if (P == 1) {
    for (int v = 0; v < nvals; ++v) out[v] = 0.0;
    for (int64_t i = 0; i < m; ++i) {
      fn(ctx, idx[i], term);
      for (int v = 0; v < nvals; ++v) out[v] += term[v];
    }
    return;
  }
  double* part = (double*)calloc((size_t)256 * nvals, sizeof(double));
  for (int l = 0; l < 256; ++l)
    for (int64_t i = l; i < m; i += 256) {
      fn(ctx, idx[i], term);
      for (int v = 0; v < nvals; ++v) part[l * nvals + v] += term[v];
    }
  tree256(part, nvals, nvals);
  for (int v = 0; v < nvals; ++v) out[v] = part[v];
  free(part);
}

/* ----------------------------------------------------------------------- EPnP -- */
typedef struct {
  const double* xy;       /* [n][2] pixels (float32 values): what the inlier test compares */
  const double* us;       /* [n][2] what the EPnP SOLVER sees: solvePnP runs undistortPoints
                           * (float32 output: (float)((u - cx) * (1 / fx))) and epnp::init_points
                           * maps that back with x * fu + uc -- a float32 round trip of the
                           * NORMALISED coordinate */
  const double* xyz;      /* [n][3] */
  double fu, fv, uc, vc;
  double c0[3];           /* centroid of the object points = control point 0 */
  double cinv[9];         /* inverse of [c1 - c0 | c2 - c0 | c3 - c0] */
  double pose[3][12];     /* candidates */
} EpnpCtx;

static void term_pw(const void* c, int32_t p, double* o) {
  const EpnpCtx* e = (const EpnpCtx*)c;
  o[0] = e->xyz[3 * p]; o[1] = e->xyz[3 * p + 1]; o[2] = e->xyz[3 * p + 2];
}
static void term_cov(const void* c, int32_t p, double* o) {
  const EpnpCtx* e = (const EpnpCtx*)c;
  const double d0 = e->xyz[3 * p] - e->c0[0], d1 = e->xyz[3 * p + 1] - e->c0[1],
               d2 = e->xyz[3 * p + 2] - e->c0[2];
  o[0] = d0 * d0; o[1] = d0 * d1; o[2] = d0 * d2; o[3] = d1 * d1; o[4] = d1 * d2; o[5] = d2 * d2;
}
static void alphas_of(const EpnpCtx* e, int32_t p, double* a) {
  const double d0 = e->xyz[3 * p] - e->c0[0], d1 = e->xyz[3 * p + 1] - e->c0[1],
               d2 = e->xyz[3 * p + 2] - e->c0[2];
  for (int j = 0; j < 3; ++j)
    a[1 + j] = e->cinv[3 * j] * d0 + e->cinv[3 * j + 1] * d1 + e->cinv[3 * j + 2] * d2;
  a[0] = 1.0 - a[1] - a[2] - a[3];
}
/* the 40 sums behind M^T M: for i <= j: a_i a_j * {1, du, dv, du^2 + dv^2} */
static void term_mtm(const void* c, int32_t p, double* o) {
  const EpnpCtx* e = (const EpnpCtx*)c;
  double a[4];
  alphas_of(e, p, a);
  const double du = e->uc - e->us[2 * p], dv = e->vc - e->us[2 * p + 1];
  const double dd = du * du + dv * dv;
  int v = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = i; j < 4; ++j) {
      const double w = a[i] * a[j];
      o[v] = w; o[v + 1] = w * du; o[v + 2] = w * dv; o[v + 3] = w * dd;
      v += 4;
    }
}
/* reprojection error (pixels, not squared) of the three candidates */
static void term_rep(const void* c, int32_t p, double* o) {
  const EpnpCtx* e = (const EpnpCtx*)c;
  const double* X = e->xyz + 3 * p;
  for (int q = 0; q < 3; ++q) {
    const double* P = e->pose[q];
    const double Xc = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[9];
    const double Yc = P[3] * X[0] + P[4] * X[1] + P[5] * X[2] + P[10];
    const double iz = 1.0 / (P[6] * X[0] + P[7] * X[1] + P[8] * X[2] + P[11]);
    const double ue = e->uc + e->fu * Xc * iz, ve = e->vc + e->fv * Yc * iz;
    const double du = e->us[2 * p] - ue, dv = e->us[2 * p + 1] - ve;
    o[q] = sqrt(du * du + dv * dv);
  }
}

static const int EP_PAIR[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};

static void gauss_newton(double L[6][10], const double* rho, double* be) {
  for (int it = 0; it < 5; ++it) {
    double A[6][5], b[6], x[5];
    for (int i = 0; i < 6; ++i) {
      const double* r = L[i];
      A[i][0] = 2.0 * r[0] * be[0] + r[1] * be[1] + r[3] * be[2] + r[6] * be[3];
      A[i][1] = r[1] * be[0] + 2.0 * r[2] * be[1] + r[4] * be[2] + r[7] * be[3];
      A[i][2] = r[3] * be[0] + r[4] * be[1] + 2.0 * r[5] * be[2] + r[8] * be[3];
      A[i][3] = r[6] * be[0] + r[7] * be[1] + r[8] * be[2] + 2.0 * r[9] * be[3];
      b[i] = rho[i] - (r[0] * be[0] * be[0] + r[1] * be[0] * be[1] + r[2] * be[1] * be[1] +
                       r[3] * be[0] * be[2] + r[4] * be[1] * be[2] + r[5] * be[2] * be[2] +
                       r[6] * be[0] * be[3] + r[7] * be[1] * be[3] + r[8] * be[2] * be[3] +
                       r[9] * be[3] * be[3]);
    }
    if (qr_solve6(4, A, b, x)) return;
    for (int i = 0; i < 4; ++i) be[i] += x[i];
  }
}

/* EPnP over the correspondences idx[0..m) (m >= 4 distinct points). P selects the summation
 * order. Returns 0 and pose = [R row-major | t], or 1 when the configuration is degenerate. */
static int epnp(EpnpCtx* e, const int32_t* idx, int64_t m, int P, double* pose) {
  const double n = (double)m;
  double s3[3], s6[6], s40[40];
  /* control points */
  canon_sum(term_pw, e, idx, m, P, 3, s3);
  for (int j = 0; j < 3; ++j) e->c0[j] = s3[j] / n;
  canon_sum(term_cov, e, idx, m, P, 6, s6);
  double cov[9] = {s6[0], s6[1], s6[2], s6[1], s6[3], s6[4], s6[2], s6[4], s6[5]};
  double A3[9], V3[9];
  memcpy(A3, cov, sizeof A3);
  jacobi_sym(3, A3, V3);
  double cc[9];                         /* column j = control point (j + 1) - control point 0 */
  {
    /* principal directions by decreasing eigenvalue (ties: lower column), each signed so
     * that its component of largest magnitude (first one on ties) is positive */
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2 - a; ++b)
        if (A3[4 * ord[b + 1]] > A3[4 * ord[b]]) { const int t = ord[b]; ord[b] = ord[b + 1]; ord[b + 1] = t; }
    for (int j = 0; j < 3; ++j) {
      const int col = ord[j];
      const double ev = A3[4 * col];
      double k = sqrt((ev > 0.0 ? ev : 0.0) / n);
      int im = 0;
      for (int i = 1; i < 3; ++i)
        if (fabs(V3[3 * i + col]) > fabs(V3[3 * im + col])) im = i;
      if (V3[3 * im + col] < 0.0) k = -k;
      for (int i = 0; i < 3; ++i) cc[3 * i + j] = k * V3[3 * i + col];
    }
  }
  if (inv3(cc, e->cinv)) return 1;
  /* M^T M */
  canon_sum(term_mtm, e, idx, m, P, 40, s40);
  double M[144], V[144];
  {
    int v = 0;
    const double fu = e->fu, fv = e->fv;
    for (int i = 0; i < 4; ++i)
      for (int j = i; j < 4; ++j) {
        const double S0 = s40[v], S1 = s40[v + 1], S2 = s40[v + 2], S3 = s40[v + 3];
        v += 4;
        const double blk[9] = {fu * fu * S0, 0.0, fu * S1, 0.0, fv * fv * S0, fv * S2,
                               fu * S1, fv * S2, S3};
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) {
            M[(3 * i + r) * 12 + 3 * j + c] = blk[3 * r + c];
            M[(3 * j + c) * 12 + 3 * i + r] = blk[3 * r + c];
          }
      }
  }
  jacobi12_rr(M, V);
  /* the four eigenvectors of smallest eigenvalue, smallest first (ties: lower column) */
  int order[4];
  {
    int used[12] = {0};
    for (int k = 0; k < 4; ++k) {
      int best = -1;
      for (int j = 0; j < 12; ++j)
        if (!used[j] && (best < 0 || M[13 * j] < M[13 * best])) best = j;
      used[best] = 1; order[k] = best;
    }
  }
  double v4[4][12];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 12; ++i) v4[k][i] = V[i * 12 + order[k]];
  /* distances between control points */
  double L[6][10], rho[6];
  {
    double cw[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   /* relative to c0 */
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) cw[1 + j][i] = cc[3 * i + j];
    for (int r = 0; r < 6; ++r) {
      const int a = EP_PAIR[r][0], b = EP_PAIR[r][1];
      double dv[4][3];
      for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 3; ++i) dv[k][i] = v4[k][3 * a + i] - v4[k][3 * b + i];
#define EP_DOT(x, y) (dv[x][0] * dv[y][0] + dv[x][1] * dv[y][1] + dv[x][2] * dv[y][2])
      L[r][0] = EP_DOT(0, 0); L[r][1] = 2.0 * EP_DOT(0, 1); L[r][2] = EP_DOT(1, 1);
      L[r][3] = 2.0 * EP_DOT(0, 2); L[r][4] = 2.0 * EP_DOT(1, 2); L[r][5] = EP_DOT(2, 2);
      L[r][6] = 2.0 * EP_DOT(0, 3); L[r][7] = 2.0 * EP_DOT(1, 3); L[r][8] = 2.0 * EP_DOT(2, 3);
      L[r][9] = EP_DOT(3, 3);
#undef EP_DOT
      const double d0 = cw[a][0] - cw[b][0], d1 = cw[a][1] - cw[b][1], d2 = cw[a][2] - cw[b][2];
      rho[r] = d0 * d0 + d1 * d1 + d2 * d2;
    }
  }
  /* the three closed-form initialisations + Gauss-Newton */
  double betas[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  int ok[3] = {1, 1, 1};
  {
    double A[6][5], b[6], x[5];
    static const int C1[4] = {0, 1, 3, 6};
    for (int i = 0; i < 6; ++i) { for (int c = 0; c < 4; ++c) A[i][c] = L[i][C1[c]]; b[i] = rho[i]; }
    if (qr_solve6(4, A, b, x)) ok[0] = 0;
    else {
      double* be = betas[0];
      if (x[0] < 0.0) { be[0] = sqrt(-x[0]); be[1] = -x[1] / be[0]; be[2] = -x[2] / be[0]; be[3] = -x[3] / be[0]; }
      else { be[0] = sqrt(x[0]); be[1] = x[1] / be[0]; be[2] = x[2] / be[0]; be[3] = x[3] / be[0]; }
    }
    for (int i = 0; i < 6; ++i) { for (int c = 0; c < 3; ++c) A[i][c] = L[i][c]; b[i] = rho[i]; }
    if (qr_solve6(3, A, b, x)) ok[1] = 0;
    else {
      double* be = betas[1];
      if (x[0] < 0.0) { be[0] = sqrt(-x[0]); be[1] = x[2] < 0.0 ? sqrt(-x[2]) : 0.0; }
      else { be[0] = sqrt(x[0]); be[1] = x[2] > 0.0 ? sqrt(x[2]) : 0.0; }
      if (x[1] < 0.0) be[0] = -be[0];
      be[2] = 0.0; be[3] = 0.0;
    }
    for (int i = 0; i < 6; ++i) { for (int c = 0; c < 5; ++c) A[i][c] = L[i][c]; b[i] = rho[i]; }
    if (qr_solve6(5, A, b, x)) ok[2] = 0;
    else {
      double* be = betas[2];
      if (x[0] < 0.0) { be[0] = sqrt(-x[0]); be[1] = x[2] < 0.0 ? sqrt(-x[2]) : 0.0; }
      else { be[0] = sqrt(x[0]); be[1] = x[2] > 0.0 ? sqrt(x[2]) : 0.0; }
      if (x[1] < 0.0) be[0] = -be[0];
      be[2] = x[3] / be[0]; be[3] = 0.0;
    }
  }
  const double* Xf = e->xyz + 3 * idx[0];      /* the first correspondence fixes the sign */
  const double df[3] = {Xf[0] - e->c0[0], Xf[1] - e->c0[1], Xf[2] - e->c0[2]};
  for (int q = 0; q < 3; ++q) {
    double* pq = e->pose[q];
    if (ok[q]) gauss_newton(L, rho, betas[q]);
    /* control points in the camera frame; pc = cc0 + B (pw - c0), B = D cinv */
    double ccs[4][3];
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < 3; ++i)
        ccs[j][i] = betas[q][0] * v4[0][3 * j + i] + betas[q][1] * v4[1][3 * j + i] +
                    betas[q][2] * v4[2][3 * j + i] + betas[q][3] * v4[3][3 * j + i];
    double B[9];
    for (int i = 0; i < 3; ++i)
      for (int c = 0; c < 3; ++c)
        B[3 * i + c] = (ccs[1][i] - ccs[0][i]) * e->cinv[c] + (ccs[2][i] - ccs[0][i]) * e->cinv[3 + c] +
                       (ccs[3][i] - ccs[0][i]) * e->cinv[6 + c];
    double pc0[3] = {ccs[0][0], ccs[0][1], ccs[0][2]};
    const double zf = pc0[2] + (B[6] * df[0] + B[7] * df[1] + B[8] * df[2]);
    if (zf < 0.0) {
      for (int i = 0; i < 9; ++i) B[i] = -B[i];
      for (int i = 0; i < 3; ++i) pc0[i] = -pc0[i];
    }
    double ABt[9], U[9], Vt[9];
    for (int i = 0; i < 3; ++i)
      for (int c = 0; c < 3; ++c)
        ABt[3 * i + c] = B[3 * i] * cov[c] + B[3 * i + 1] * cov[3 + c] + B[3 * i + 2] * cov[6 + c];
    svd3(ABt, U, Vt);
    for (int i = 0; i < 3; ++i)
      for (int c = 0; c < 3; ++c)
        pq[3 * i + c] = U[3 * i] * Vt[3 * c] + U[3 * i + 1] * Vt[3 * c + 1] + U[3 * i + 2] * Vt[3 * c + 2];
    const double det = pq[0] * (pq[4] * pq[8] - pq[5] * pq[7]) - pq[1] * (pq[3] * pq[8] - pq[5] * pq[6]) +
                       pq[2] * (pq[3] * pq[7] - pq[4] * pq[6]);
    if (det < 0.0) { pq[6] = -pq[6]; pq[7] = -pq[7]; pq[8] = -pq[8]; }
    for (int i = 0; i < 3; ++i)
      pq[9 + i] = pc0[i] - (pq[3 * i] * e->c0[0] + pq[3 * i + 1] * e->c0[1] + pq[3 * i + 2] * e->c0[2]);
    for (int i = 0; i < 12; ++i)
      if (!(pq[i] == pq[i]) || !ok[q]) { ok[q] = 0; break; }
    if (!ok[q]) for (int i = 0; i < 12; ++i) pq[i] = i % 4 == 0 && i < 9 ? 1.0 : 0.0;
  }
  double rep[3];
  canon_sum(term_rep, e, idx, m, P, 3, rep);
  int best = -1;
  for (int q = 0; q < 3; ++q) {
    if (!ok[q] || !(rep[q] == rep[q])) continue;
    if (best < 0 || rep[q] < rep[best]) best = q;
  }
  if (best < 0) return 1;
  memcpy(pose, e->pose[best], 12 * sizeof(double));
  return 0;
}

/* ----------------------------------------------------- RANSAC around it -- */
static int is_inlier(const double* pose, const EpnpCtx* e, int32_t p, float t2) {
  const double* X = e->xyz + 3 * p;
  const double Xc = pose[0] * X[0] + pose[1] * X[1] + pose[2] * X[2] + pose[9];
  const double Yc = pose[3] * X[0] + pose[4] * X[1] + pose[5] * X[2] + pose[10];
  const double Zc = pose[6] * X[0] + pose[7] * X[1] + pose[8] * X[2] + pose[11];
  const double iz = Zc != 0.0 ? 1.0 / Zc : 1.0;
  const float up = (float)((Xc * iz) * e->fu + e->uc), vp = (float)((Yc * iz) * e->fv + e->vc);
  const float dx = (float)e->xy[2 * p] - up, dy = (float)e->xy[2 * p + 1] - vp;
  const float err = (float)((double)dx * (double)dx + (double)dy * (double)dy);
  return err <= t2;
}

/* w^5 rounded ONCE (double-double products by fma, then one addition): what a correctly
 * rounded pow(w, 5) returns -- OpenCV's RANSACUpdateNumIters calls std::pow(1 - ep,
 * modelPoints), and glibc's pow is correctly rounded in all but astronomically rare cases.
 * (w * w) * (w * w) * w, four roundings, can be an ulp off. 0 <= w <= 1. */
static double pow5_rn(double w) {
  const double h2 = w * w, l2 = fma(w, w, -h2);                 /* w^2 = h2 + l2 exactly */
  const double h4 = h2 * h2;
  const double l4 = fma(h2, h2, -h4) + 2.0 * (h2 * l2);         /* w^4 ~ h4 + l4 */
  const double h5 = h4 * w;
  const double l5 = fma(h4, w, -h5) + l4 * w;                   /* w^5 ~ h5 + l5 */
  return h5 + l5;
}

/* the float32 round trip of the normalised image coordinate (see EpnpCtx.us) */
static double us_of(double u_f32, double c, double f) {
  const double inv = 1.0 / f;
  const float xn = (float)((u_f32 - c) * inv);
  return (double)xn * f + c;
}

static int update_niters(double p, double ep, int max_iters) {
  if (p < 0.0) p = 0.0;
  if (p > 1.0) p = 1.0;
  if (ep < 0.0) ep = 0.0;
  if (ep > 1.0) ep = 1.0;
  double num = 1.0 - p;
  if (num < DBL_MIN) num = DBL_MIN;
  const double w = 1.0 - ep;
  double denom = 1.0 - pow5_rn(w);
  if (denom < DBL_MIN) return 0;
  num = log(num);
  denom = log(denom);
  if (denom >= 0.0 || -num >= (double)max_iters * (-denom)) return max_iters;
  return (int)nearbyint(num / denom);
}

/* Returns 1 and pose_out = [R | t] / inlier_mask, or 0 ("pose_est_success" false).
 * info (optional, int32[4]): index of the best minimal set, its inlier count, the final
 * iteration bound, the number of iterations actually evaluated. */
int epnp_ref_solve_pnp_ransac(const double* xy_in, const double* xyz_in, int64_t n,
                              const double* K, int max_iters, double reproj_err,
                              double confidence, double* pose_out, uint8_t* inlier_mask,
                              int32_t* info) {
  if (info) { info[0] = -1; info[1] = 0; info[2] = max_iters; info[3] = 0; }
  if (n < EP_MODEL_POINTS || max_iters < 1) return 0;
  double* xy = (double*)malloc((size_t)n * 2 * sizeof(double));
  double* xyz = (double*)malloc((size_t)n * 3 * sizeof(double));
  int32_t* idx = (int32_t*)malloc((size_t)n * sizeof(int32_t));
  double* us = (double*)malloc((size_t)n * 2 * sizeof(double));
  for (int64_t i = 0; i < 2 * n; ++i) xy[i] = (double)(float)xy_in[i];
  for (int64_t i = 0; i < 3 * n; ++i) xyz[i] = (double)(float)xyz_in[i];
  for (int64_t i = 0; i < n; ++i) {
    us[2 * i] = us_of(xy[2 * i], K[2], K[0]);
    us[2 * i + 1] = us_of(xy[2 * i + 1], K[5], K[4]);
  }
  EpnpCtx e;
  memset(&e, 0, sizeof e);
  e.xy = xy; e.us = us; e.xyz = xyz; e.fu = K[0]; e.fv = K[4]; e.uc = K[2]; e.vc = K[5];
  const float t2 = (float)(reproj_err * reproj_err);
  uint64_t rng = 0xffffffffffffffffull;
  int niters = max_iters, best_count = 0, best_it = -1, it = 0;
  double best_pose[12], pose[12];
  if (n == EP_MODEL_POINTS) {            /* RANSACPointSetRegistrator::run: one kernel call */
    for (int i = 0; i < 5; ++i) idx[i] = i;
    if (!epnp(&e, idx, 5, 1, best_pose)) { best_it = 0; best_count = 5; }
    for (int64_t i = 0; i < n; ++i) inlier_mask[i] = best_it >= 0;
    niters = 0;
  }
  for (; it < niters; ++it) {
    int32_t s[EP_MODEL_POINTS];
    for (int i = 0; i < EP_MODEL_POINTS;) {
      const int32_t c = (int32_t)(cvrng_next(&rng) % (uint32_t)n);
      int j = 0;
      for (; j < i; ++j) if (s[j] == c) break;
      if (j == i) s[i++] = c;
    }
    if (epnp(&e, s, EP_MODEL_POINTS, 1, pose)) continue;
    int count = 0;
    for (int64_t p = 0; p < n; ++p) count += is_inlier(pose, &e, (int32_t)p, t2);
    if (count > (best_count > EP_MODEL_POINTS - 1 ? best_count : EP_MODEL_POINTS - 1)) {
      best_count = count; best_it = it;
      memcpy(best_pose, pose, sizeof pose);
      niters = update_niters(confidence, (double)(n - count) / (double)n, niters);
    }
  }
  int ok = 0;
  if (best_it >= 0) {
    int64_t m = 0;
    for (int64_t p = 0; p < n; ++p) {
      const int in = n == EP_MODEL_POINTS ? 1 : is_inlier(best_pose, &e, (int32_t)p, t2);
      inlier_mask[p] = (uint8_t)in;
      if (in) idx[m++] = (int32_t)p;
    }
    ok = !epnp(&e, idx, m, 256, pose_out);
  }
  if (!ok) for (int64_t p = 0; p < n; ++p) inlier_mask[p] = 0;
  if (info) { info[0] = best_it; info[1] = best_count; info[2] = niters; info[3] = it; }
  free(xy); free(us); free(xyz); free(idx);
  return ok;
}

/* test hooks */
double epnp_ref_pow5(double w) { return pow5_rn(w); }
double epnp_ref_us_of(double u_f32, double c, double f) { return us_of(u_f32, c, f); }

/* EPnP alone over all n correspondences (tests): order = 1 or 256 */
int epnp_ref_epnp(const double* xy, const double* xyz, int64_t n, const double* K, int order,
                  double* pose_out) {
  EpnpCtx e;
  memset(&e, 0, sizeof e);
  e.xy = xy; e.us = xy; e.xyz = xyz; e.fu = K[0]; e.fv = K[4]; e.uc = K[2]; e.vc = K[5];
  int32_t* idx = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  for (int64_t i = 0; i < n; ++i) idx[i] = (int32_t)i;
  const int r = n >= 4 ? epnp(&e, idx, n, order, pose_out) : 1;
  free(idx);
  return r;
}

void epnp_ref_jacobi(int n, double* A, double* V) {
  if (n == 12) jacobi12_rr(A, V); else jacobi_sym(n, A, V);
}
uint32_t epnp_ref_rng_next(uint64_t* st) { return cvrng_next(st); }
