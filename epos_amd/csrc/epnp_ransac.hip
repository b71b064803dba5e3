// cv2.solvePnPRansac(..., flags=cv2.SOLVEPNP_EPNP) for gfx950 -- the reference's alternative
// fitting method (scripts/infer.py:505-528: iterationsCount = max_fitting_iterations,
// reprojectionError = inlier_thresh, confidence = 0.99, one instance per object). OpenCV
// (README.md:29, opencv=3.4.2) is not vendored; the algorithm is restated in
// include/epos_hip.h and DESIGN.md "The OpenCV fitting method".
//
// Mapping to the hardware:
//   * cvr_samples: the sampler is cv::RNG, a SEQUENTIAL generator (multiply-with-carry, an
//     index that repeats inside a 5-set is re-drawn), so one lane per slot walks it once and
//     writes the whole table of minimal sets; everything after that is parallel.
//   * cvr_hypotheses: ONE MINIMAL SET PER WAVEFRONT, grid = (iterations / 4, slots). The 64
//     lanes run the wave-uniform part of EPnP redundantly (control points, the 6 x 10 system,
//     the three beta initialisations + Gauss-Newton, absolute orientation); the 12 x 12
//     symmetric eigenproblem -- four fifths of the flops -- is a Jacobi iteration held in LDS
//     in round-robin order: the six disjoint rotations of a round are applied at once, every
//     lane updating its three entries of the matrix; the inliers of the
//     resulting pose are counted with the lanes strided over the slot's correspondences.
//   * cvr_select_fit: one workgroup per slot. RANSAC's "keep the best, shrink the iteration
//     bound" is replayed over the table of inlier counts in iteration order by one lane
//     (identical to running the loop sequentially: sets drawn after the bound are ignored);
//     the inliers of the winner are compacted in index order and EPnP runs once more over all
//     of them, its sums over correspondences as 256 strided partials + butterfly.
// fp64 with + - * / sqrt only (log() once per accepted set, in the bound), float32 exactly
// where OpenCV evaluates the reprojection error in float32; -ffp-contract=off. The results
// are the bits of a scalar evaluation in the same canonical order.
#include <float.h>

#include "common.h"

namespace epos {
namespace {

#ifndef EPOS_EP_SWEEPS
#define EPOS_EP_SWEEPS 12      // timing experiments only: the oracle uses 12
#endif
constexpr int EP_SWEEPS = EPOS_EP_SWEEPS;
constexpr int EP_SET = 5;            // model_points of solvePnPRansac for EPNP

struct EpCam { double fu, fv, uc, vc; };
struct EpFrame {
  double c0[3];      // centroid of the object points = control point 0
  double cov[9];     // sum (pw - c0)(pw - c0)^T
  double cc[9];      // column j = control point (j + 1) - c0
  double cinv[9];    // cc^-1
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ double f32r(double v) { return static_cast<double>(static_cast<float>(v)); }

// ---------------------------------------------------------- small dense algebra --
// cyclic Jacobi, 3 x 3, in registers (all indices static after unrolling)
__device__ void jacobi3(double* A, double* V) {
#pragma unroll
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < EP_SWEEPS; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        const double app = A[p * 3 + p], aqq = A[q * 3 + q];
        if (fabs(apq) <= 8.673617379884035e-19 * (fabs(app) + fabs(aqq))) {
          A[p * 3 + q] = 0.0; A[q * 3 + p] = 0.0;
          continue;
        }
        const double theta = (aqq - app) / (2.0 * apq);
        double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
        if (theta < 0.0) t = -t;
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (k != p && k != q) {
            const double akp = A[k * 3 + p], akq = A[k * 3 + q];
            const double np_ = c * akp - s * akq, nq_ = s * akp + c * akq;
            A[k * 3 + p] = np_; A[p * 3 + k] = np_;
            A[k * 3 + q] = nq_; A[q * 3 + k] = nq_;
          }
          const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
        A[p * 3 + p] = app - t * apq;
        A[q * 3 + q] = aqq + t * apq;
        A[p * 3 + q] = 0.0; A[q * 3 + p] = 0.0;
        rotated = true;
      }
    if (!rotated) break;
  }
}

// The 12 x 12 problem in LDS, by ONE wavefront, in ROUND-ROBIN order: a sweep is 11 rounds of
// 6 disjoint pairs (round r, pair m: (r, 11) for m = 0, else ((r + m) mod 11, (r - m) mod 11);
// the first index plays "p", the second "q" -- a rotation gives the same bits with the roles
// swapped, so no sorting). The six rotations of a round commute; the oracle applies them one
// after the other in the order m = 0..5, which fixes the rounding of an entry (i, j) whose row
// belongs to pair a and whose column to pair b: first the rotation of min(a, b) along its
// index, then the other one. Here lanes 0..5 compute the six angles, then every lane
// evaluates that expression for its three entries of A and of V from the values of BEFORE
// the round. A lane owns the same (pair, role) x (pair, role) slots in every round -- what
// changes with the round is only which matrix indices the pairs stand for -- so everything
// about the slot (which formula, which rotation parameters) is decided once, outside the loops.
__device__ __forceinline__ int rr_first(int m, int r) { const int v = r + m; return m == 0 ? r : (v >= 11 ? v - 11 : v); }
__device__ __forceinline__ int rr_second(int m, int r) { const int v = r - m; return m == 0 ? 11 : (v < 0 ? v + 11 : v); }
__device__ __forceinline__ void jacobi12_wave(double* A, double* V, double* R /*[6][4]*/, int lane) {
  for (int e = lane; e < 144; e += 64) V[e] = (e % 13 == 0) ? 1.0 : 0.0;
  int ma[3], mb[3], vk[3], mx[3];
  bool ra[3], rb[3], swp[3], same[3], diag[3], rx[3], live[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    int e = lane + 64 * u;
    live[u] = e < 144;
    e = live[u] ? e : 0;
    int a = e / 24, b = (e % 12) / 2;
    bool qa = (e / 12) % 2 != 0, qb = e % 2 != 0;
    vk[u] = e / 12; mx[u] = b; rx[u] = qb;                   // V[k][(pair, role)]
    swp[u] = a > b;
    if (swp[u]) { const int t = a; a = b; b = t; const bool tq = qa; qa = qb; qb = tq; }
    ma[u] = a; mb[u] = b; ra[u] = qa; rb[u] = qb;
    same[u] = a == b; diag[u] = a == b && qa == qb;
  }
  wave_sync();
  for (int sweep = 0; sweep < EP_SWEEPS; ++sweep) {
    bool rotated = false;
    for (int r = 0; r < 11; ++r) {
      if (lane < 6) {                      // flag 0: untouched, 1: a_pq := 0, 2: rotate
        const int m = lane;
        const int p = rr_first(m, r), q = rr_second(m, r);
        const double apq = A[p * 12 + q];
        double c = 1.0, s = 0.0, t = 0.0, flag = 0.0;
        if (apq != 0.0) {
          const double app = A[p * 12 + p], aqq = A[q * 12 + q];
          if (fabs(apq) <= 8.673617379884035e-19 * (fabs(app) + fabs(aqq))) flag = 1.0;
          else {
            const double theta = (aqq - app) / (2.0 * apq);
            t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
            if (theta < 0.0) t = -t;
            c = 1.0 / sqrt(t * t + 1.0); s = t * c;
            flag = 2.0;
          }
        }
        R[4 * m] = c; R[4 * m + 1] = s; R[4 * m + 2] = t; R[4 * m + 3] = flag;
      }
      wave_sync();
      bool touch = false, rot = false;
#pragma unroll
      for (int m = 0; m < 6; ++m) { const double f = R[4 * m + 3]; touch = touch || f != 0.0; rot = rot || f == 2.0; }
      if (touch) {                                         // wave-uniform
        double na[3], nv[3];
        int wa[3], wv[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int ip = rr_first(ma[u], r), iq = rr_second(ma[u], r);
          const int jp = rr_first(mb[u], r), jq = rr_second(mb[u], r);
          const int xp = rr_first(mx[u], r), xq = rr_second(mx[u], r);
          const double app_ = A[ip * 12 + jp], aqp_ = A[iq * 12 + jp];
          const double apq_ = A[ip * 12 + jq], aqq_ = A[iq * 12 + jq];
          const double vp_ = V[vk[u] * 12 + xp], vq_ = V[vk[u] * 12 + xq];
          const double ca = R[4 * ma[u]], sa = R[4 * ma[u] + 1], ta = R[4 * ma[u] + 2], fa = R[4 * ma[u] + 3];
          const double cb = R[4 * mb[u]], sb = R[4 * mb[u] + 1], fb = R[4 * mb[u] + 3];
          // V[k][x]: the rotation of x's pair (= the slot's column pair before the swap)
          const double cv = swp[u] ? ca : cb, sv = swp[u] ? sa : sb, fv = swp[u] ? fa : fb;
          const double vrot = rx[u] ? sv * vp_ + cv * vq_ : cv * vp_ - sv * vq_;
          nv[u] = fv == 2.0 ? vrot : (rx[u] ? vq_ : vp_);
          wv[u] = vk[u] * 12 + (rx[u] ? xq : xp);
          // A, entry in two different pairs: rotation a along i, then rotation b along j
          const double old_p = ra[u] ? aqp_ : app_, old_q = ra[u] ? aqq_ : apq_;
          const double r1p = ra[u] ? sa * app_ + ca * aqp_ : ca * app_ - sa * aqp_;
          const double r1q = ra[u] ? sa * apq_ + ca * aqq_ : ca * apq_ - sa * aqq_;
          const double n1p = fa == 2.0 ? r1p : old_p, n1q = fa == 2.0 ? r1q : old_q;
          const double r2 = rb[u] ? sb * n1p + cb * n1q : cb * n1p - sb * n1q;
          const double off = fb == 2.0 ? r2 : (rb[u] ? n1q : n1p);
          // A, entry inside one pair's 2 x 2 block (then ip = jp = p, iq = jq = q)
          const double drot = ra[u] ? aqq_ + ta * apq_ : app_ - ta * apq_;
          const double blk_diag = fa == 2.0 ? drot : (ra[u] ? aqq_ : app_);
          const double blk_off = fa == 0.0 ? apq_ : 0.0;
          na[u] = same[u] ? (diag[u] ? blk_diag : blk_off) : off;
          const int i = ra[u] ? iq : ip, j = rb[u] ? jq : jp;
          wa[u] = swp[u] ? j * 12 + i : i * 12 + j;
        }
        wave_sync();                                       // every read before any write
#pragma unroll
        for (int u = 0; u < 3; ++u)
          if (live[u]) { A[wa[u]] = na[u]; V[wv[u]] = nv[u]; }
        rotated = rotated || rot;
      }
      wave_sync();
    }
    if (!rotated) break;
  }
}

// min ||A x - b|| for a 6 x K system by Householder QR (A row-major, leading dimension 5)
template <int K>
__device__ bool qr_solve6(double (*A)[5], double* b, double* x) {
#pragma unroll
  for (int j = 0; j < K; ++j) {
    double nrm2 = 0.0;
#pragma unroll
    for (int i = j; i < 6; ++i) nrm2 += A[i][j] * A[i][j];
    if (!(nrm2 > 0.0)) return true;
    const double alpha = A[j][j] > 0.0 ? -sqrt(nrm2) : sqrt(nrm2);
    double v[6];
#pragma unroll
    for (int i = j; i < 6; ++i) v[i] = A[i][j];
    v[j] = v[j] - alpha;
    double vn2 = 0.0;
#pragma unroll
    for (int i = j; i < 6; ++i) vn2 += v[i] * v[i];
    if (!(vn2 > 0.0)) return true;
#pragma unroll
    for (int c = j + 1; c < K; ++c) {
      double d = 0.0;
#pragma unroll
      for (int i = j; i < 6; ++i) d += v[i] * A[i][c];
      const double f = 2.0 * d / vn2;
#pragma unroll
      for (int i = j; i < 6; ++i) A[i][c] = A[i][c] - f * v[i];
    }
    double d = 0.0;
#pragma unroll
    for (int i = j; i < 6; ++i) d += v[i] * b[i];
    const double f = 2.0 * d / vn2;
#pragma unroll
    for (int i = j; i < 6; ++i) b[i] = b[i] - f * v[i];
    A[j][j] = alpha;
  }
#pragma unroll
  for (int j = K - 1; j >= 0; --j) {
    double s = b[j];
#pragma unroll
    for (int c = j + 1; c < K; ++c) s -= A[j][c] * x[c];
    x[j] = s / A[j][j];
  }
  return false;
}

// M = U diag(S) V^T, 3 x 3, one-sided Jacobi; a vanishing direction of U is completed
__device__ void svd3(const double* M, double* U, double* V) {
  double G[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { G[i] = M[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < EP_SWEEPS; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int p = r == 2 ? 1 : 0, q = r == 0 ? 1 : 2;
      const double al = G[p] * G[p] + G[3 + p] * G[3 + p] + G[6 + p] * G[6 + p];
      const double be = G[q] * G[q] + G[3 + q] * G[3 + q] + G[6 + q] * G[6 + q];
      const double ga = G[p] * G[q] + G[3 + p] * G[3 + q] + G[6 + p] * G[6 + q];
      if (ga == 0.0 || ga * ga <= 1e-34 * (al * be)) continue;
      const double zeta = (be - al) / (2.0 * ga);
      double t = 1.0 / (fabs(zeta) + sqrt(zeta * zeta + 1.0));
      if (zeta < 0.0) t = -t;
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double gp = G[3 * k + p], gq = G[3 * k + q];
        G[3 * k + p] = c * gp - s * gq; G[3 * k + q] = s * gp + c * gq;
        const double vp = V[3 * k + p], vq = V[3 * k + q];
        V[3 * k + p] = c * vp - s * vq; V[3 * k + q] = s * vp + c * vq;
      }
      rotated = true;
    }
    if (!rotated) break;
  }
  double S[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) S[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
  int jmin = 0;
  if (S[1] < S[jmin]) jmin = 1;
  if (S[2] < S[jmin]) jmin = 2;
  double smax = S[0] > S[1] ? S[0] : S[1];
  if (S[2] > smax) smax = S[2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double inv = 1.0 / S[j];
    U[j] = G[j] * inv; U[3 + j] = G[3 + j] * inv; U[6 + j] = G[6 + j] * inv;
  }
  if (!(S[jmin] > 1e-12 * smax)) {
#pragma unroll
    for (int jm = 0; jm < 3; ++jm)
      if (jm == jmin) {
        const int a = (jm + 1) % 3, b = (jm + 2) % 3;
        U[jm] = U[3 + a] * U[6 + b] - U[6 + a] * U[3 + b];
        U[3 + jm] = U[6 + a] * U[b] - U[a] * U[6 + b];
        U[6 + jm] = U[a] * U[3 + b] - U[3 + a] * U[b];
      }
  }
}

__device__ bool inv3(const double* m, double* inv) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8],
               c02 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  if (!(fabs(det) > 0.0)) return true;
  const double id = 1.0 / det;
  inv[0] = c00 * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  inv[3] = c01 * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  inv[6] = c02 * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return false;
}

// ------------------------------------------------------------------------ EPnP --
// per-correspondence terms of the sums (X, x already rounded to float32 values)
__device__ __forceinline__ void term_cov(const EpFrame& f, const double* X, double* o) {
  const double d0 = X[0] - f.c0[0], d1 = X[1] - f.c0[1], d2 = X[2] - f.c0[2];
  o[0] = d0 * d0; o[1] = d0 * d1; o[2] = d0 * d2; o[3] = d1 * d1; o[4] = d1 * d2; o[5] = d2 * d2;
}
__device__ __forceinline__ void term_mtm(const EpFrame& f, const EpCam& cam, const double* X,
                                         const double* x, double* o) {
  const double d0 = X[0] - f.c0[0], d1 = X[1] - f.c0[1], d2 = X[2] - f.c0[2];
  double a[4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
    a[1 + j] = f.cinv[3 * j] * d0 + f.cinv[3 * j + 1] * d1 + f.cinv[3 * j + 2] * d2;
  a[0] = 1.0 - a[1] - a[2] - a[3];
  const double du = cam.uc - x[0], dv = cam.vc - x[1];
  const double dd = du * du + dv * dv;
  int v = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = i; j < 4; ++j) {
      const double w = a[i] * a[j];
      o[v] = w; o[v + 1] = w * du; o[v + 2] = w * dv; o[v + 3] = w * dd;
      v += 4;
    }
}
__device__ __forceinline__ void term_rep(const double* poses /*[3][12]*/, const EpCam& cam,
                                         const double* X, const double* x, double* o) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const double* P = poses + 12 * q;
    const double Xc = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[9];
    const double Yc = P[3] * X[0] + P[4] * X[1] + P[5] * X[2] + P[10];
    const double iz = 1.0 / (P[6] * X[0] + P[7] * X[1] + P[8] * X[2] + P[11]);
    const double ue = cam.uc + cam.fu * Xc * iz, ve = cam.vc + cam.fv * Yc * iz;
    const double du = x[0] - ue, dv = x[1] - ve;
    o[q] = sqrt(du * du + dv * dv);
  }
}

// centroid sums + covariance sums -> control points. Returns true when degenerate.
__device__ bool ep_control_points(const double* s6, double n, EpFrame& f) {
  const double cov[9] = {s6[0], s6[1], s6[2], s6[1], s6[3], s6[4], s6[2], s6[4], s6[5]};
  double A3[9], V3[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { f.cov[i] = cov[i]; A3[i] = cov[i]; }
  jacobi3(A3, V3);
  int ord[3] = {0, 1, 2};
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2 - a; ++b)
      if (A3[4 * ord[b + 1]] > A3[4 * ord[b]]) { const int t = ord[b]; ord[b] = ord[b + 1]; ord[b + 1] = t; }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int col = ord[j];
    const double ev = A3[4 * col];
    double k = sqrt((ev > 0.0 ? ev : 0.0) / n);
    int im = 0;
#pragma unroll
    for (int i = 1; i < 3; ++i)
      if (fabs(V3[3 * i + col]) > fabs(V3[3 * im + col])) im = i;
    if (V3[3 * im + col] < 0.0) k = -k;
#pragma unroll
    for (int i = 0; i < 3; ++i) f.cc[3 * i + j] = k * V3[3 * i + col];
  }
  return inv3(f.cc, f.cinv);
}

// the 40 sums -> M^T M in LDS (one lane writes; the callers synchronise)
__device__ __forceinline__ void ep_write_mtm(const double* s40, const EpCam& cam, double* M) {
  int v = 0;
  const double fu = cam.fu, fv = cam.fv;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = i; j < 4; ++j) {
      const double S0 = s40[v], S1 = s40[v + 1], S2 = s40[v + 2], S3 = s40[v + 3];
      v += 4;
      const double blk[9] = {fu * fu * S0, 0.0, fu * S1, 0.0, fv * fv * S0, fv * S2,
                             fu * S1, fv * S2, S3};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          M[(3 * i + r) * 12 + 3 * j + c] = blk[3 * r + c];
          M[(3 * j + c) * 12 + 3 * i + r] = blk[3 * r + c];
        }
    }
}

__device__ void gauss_newton(const double (*L)[10], const double* rho, double* be) {
  for (int it = 0; it < 5; ++it) {
    double A[6][5], b[6], x[5];
#pragma unroll
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
    if (qr_solve6<4>(A, b, x)) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) be[i] += x[i];
  }
}

// After the eigen-decomposition (diagonal of A = eigenvalues, columns of V = eigenvectors, in
// LDS): the three candidate poses. Xf = the first correspondence (it fixes the sign).
// Wave- / workgroup-uniform; every calling thread computes the same values.
__device__ __forceinline__ void ep_candidates(const double* A, const double* V, const EpFrame& f,
                              const double* Xf, double* poses /*[3][12]*/, bool* ok /*[3]*/) {
  int order[4];
  {
    unsigned used = 0;
    for (int k = 0; k < 4; ++k) {
      int best = -1;
      double bv = 0.0;
      for (int j = 0; j < 12; ++j) {
        const double ev = A[13 * j];
        if (!((used >> j) & 1) && (best < 0 || ev < bv)) { best = j; bv = ev; }
      }
      used |= 1u << best; order[k] = best;
    }
  }
  double v4[4][12];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int i = 0; i < 12; ++i) v4[k][i] = V[i * 12 + order[k]];
  double L[6][10], rho[6];
  {
    double cw[4][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) cw[0][i] = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) cw[1 + j][i] = f.cc[3 * i + j];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int a = r < 3 ? 0 : (r < 5 ? 1 : 2);
      const int b = r < 3 ? r + 1 : (r < 5 ? r - 1 : 3);
      double dv[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
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
  double betas[3][4];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    ok[q] = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) betas[q][i] = 0.0;
  }
  {
    double A6[6][5], b[6], x[5];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      A6[i][0] = L[i][0]; A6[i][1] = L[i][1]; A6[i][2] = L[i][3]; A6[i][3] = L[i][6];
      b[i] = rho[i];
    }
    if (qr_solve6<4>(A6, b, x)) ok[0] = false;
    else {
      double* be = betas[0];
      if (x[0] < 0.0) { be[0] = sqrt(-x[0]); be[1] = -x[1] / be[0]; be[2] = -x[2] / be[0]; be[3] = -x[3] / be[0]; }
      else { be[0] = sqrt(x[0]); be[1] = x[1] / be[0]; be[2] = x[2] / be[0]; be[3] = x[3] / be[0]; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int c = 0; c < 3; ++c) A6[i][c] = L[i][c];
      b[i] = rho[i];
    }
    if (qr_solve6<3>(A6, b, x)) ok[1] = false;
    else {
      double* be = betas[1];
      if (x[0] < 0.0) { be[0] = sqrt(-x[0]); be[1] = x[2] < 0.0 ? sqrt(-x[2]) : 0.0; }
      else { be[0] = sqrt(x[0]); be[1] = x[2] > 0.0 ? sqrt(x[2]) : 0.0; }
      if (x[1] < 0.0) be[0] = -be[0];
      be[2] = 0.0; be[3] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int c = 0; c < 5; ++c) A6[i][c] = L[i][c];
      b[i] = rho[i];
    }
    if (qr_solve6<5>(A6, b, x)) ok[2] = false;
    else {
      double* be = betas[2];
      if (x[0] < 0.0) { be[0] = sqrt(-x[0]); be[1] = x[2] < 0.0 ? sqrt(-x[2]) : 0.0; }
      else { be[0] = sqrt(x[0]); be[1] = x[2] > 0.0 ? sqrt(x[2]) : 0.0; }
      if (x[1] < 0.0) be[0] = -be[0];
      be[2] = x[3] / be[0]; be[3] = 0.0;
    }
  }
  const double df[3] = {Xf[0] - f.c0[0], Xf[1] - f.c0[1], Xf[2] - f.c0[2]};
  for (int q = 0; q < 3; ++q) {
    double* pq = poses + 12 * q;
    double be[4] = {betas[q][0], betas[q][1], betas[q][2], betas[q][3]};
    if (ok[q]) gauss_newton(L, rho, be);
    double ccs[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        ccs[j][i] = be[0] * v4[0][3 * j + i] + be[1] * v4[1][3 * j + i] +
                    be[2] * v4[2][3 * j + i] + be[3] * v4[3][3 * j + i];
    double B[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        B[3 * i + c] = (ccs[1][i] - ccs[0][i]) * f.cinv[c] + (ccs[2][i] - ccs[0][i]) * f.cinv[3 + c] +
                       (ccs[3][i] - ccs[0][i]) * f.cinv[6 + c];
    double pc0[3] = {ccs[0][0], ccs[0][1], ccs[0][2]};
    const double zf = pc0[2] + (B[6] * df[0] + B[7] * df[1] + B[8] * df[2]);
    if (zf < 0.0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) B[i] = -B[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) pc0[i] = -pc0[i];
    }
    double ABt[9], U[9], Vm[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        ABt[3 * i + c] = B[3 * i] * f.cov[c] + B[3 * i + 1] * f.cov[3 + c] + B[3 * i + 2] * f.cov[6 + c];
    svd3(ABt, U, Vm);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        pq[3 * i + c] = U[3 * i] * Vm[3 * c] + U[3 * i + 1] * Vm[3 * c + 1] + U[3 * i + 2] * Vm[3 * c + 2];
    const double det = pq[0] * (pq[4] * pq[8] - pq[5] * pq[7]) - pq[1] * (pq[3] * pq[8] - pq[5] * pq[6]) +
                       pq[2] * (pq[3] * pq[7] - pq[4] * pq[6]);
    if (det < 0.0) { pq[6] = -pq[6]; pq[7] = -pq[7]; pq[8] = -pq[8]; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
      pq[9 + i] = pc0[i] - (pq[3 * i] * f.c0[0] + pq[3 * i + 1] * f.c0[1] + pq[3 * i + 2] * f.c0[2]);
    bool good = ok[q];
#pragma unroll
    for (int i = 0; i < 12; ++i)
      if (!(pq[i] == pq[i])) good = false;
    ok[q] = good;
    if (!good) {
#pragma unroll
      for (int i = 0; i < 12; ++i) pq[i] = (i % 4 == 0 && i < 9) ? 1.0 : 0.0;
    }
  }
}

__device__ __forceinline__ int ep_choose(const bool* ok, const double* rep) {
  int best = -1;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (!ok[q] || !(rep[q] == rep[q])) continue;
    if (best < 0 || rep[q] < rep[best]) best = q;
  }
  return best;
}

// PnPRansacCallback::computeError + findInliers: float32 projection, float32 error
__device__ __forceinline__ bool ep_is_inlier(const double* P, const EpCam& cam,
                                             const double* xy, const double* xyz, int64_t p,
                                             float t2) {
  const double X0 = f32r(xyz[3 * p]), X1 = f32r(xyz[3 * p + 1]), X2 = f32r(xyz[3 * p + 2]);
  const double Xc = P[0] * X0 + P[1] * X1 + P[2] * X2 + P[9];
  const double Yc = P[3] * X0 + P[4] * X1 + P[5] * X2 + P[10];
  const double Zc = P[6] * X0 + P[7] * X1 + P[8] * X2 + P[11];
  const double iz = Zc != 0.0 ? 1.0 / Zc : 1.0;
  const float up = static_cast<float>((Xc * iz) * cam.fu + cam.uc);
  const float vp = static_cast<float>((Yc * iz) * cam.fv + cam.vc);
  const float dx = static_cast<float>(xy[2 * p]) - up, dy = static_cast<float>(xy[2 * p + 1]) - vp;
  const float err = static_cast<float>(static_cast<double>(dx) * static_cast<double>(dx) +
                                       static_cast<double>(dy) * static_cast<double>(dy));
  return err <= t2;
}

__device__ __forceinline__ double ep_pow5_rn(double w) {        // = pow5_rn of the C definition
  const double h2 = w * w, l2 = fma(w, w, -h2);
  const double h4 = h2 * h2;
  const double l4 = fma(h2, h2, -h4) + 2.0 * (h2 * l2);
  const double h5 = h4 * w;
  const double l5 = fma(h4, w, -h5) + l4 * w;
  return h5 + l5;
}
// The image point as the EPnP SOLVER sees it (us_of of the C definition): solvePnP's
// undistortPoints writes (float)((u - cx) * (1 / fx)), epnp::init_points maps it back.
__device__ __forceinline__ double ep_us_of(double u_f32, double c, double f) {
  const double inv = 1.0 / f;
  const float xn = static_cast<float>((u_f32 - c) * inv);
  return static_cast<double>(xn) * f + c;
}

__device__ __forceinline__ int ep_update_niters(double p, double ep, int max_iters) {
  if (p < 0.0) p = 0.0;
  if (p > 1.0) p = 1.0;
  if (ep < 0.0) ep = 0.0;
  if (ep > 1.0) ep = 1.0;
  double num = 1.0 - p;
  if (num < DBL_MIN) num = DBL_MIN;
  const double w = 1.0 - ep;
  // OpenCV's RANSACUpdateNumIters takes pow(1 - ep, modelPoints) from libm. A device pow()
  // and a host pow() need not agree in the last bit, so the fifth power is formed here (and
  // in the C oracle) as what a CORRECTLY ROUNDED pow returns: double-double products by fma,
  // one final addition (round 4; the four-rounding product used before could be an ulp off).
  double denom = 1.0 - ep_pow5_rn(w);
  if (denom < DBL_MIN) return 0;
  num = log(num);
  denom = log(denom);
  if (denom >= 0.0 || -num >= static_cast<double>(max_iters) * (-denom)) return max_iters;
  return static_cast<int>(rint(num / denom));
}

// ------------------------------------------------------------------- kernels --
struct CvrWork {
  int min_points;      // slots below this size are treated as empty
  int32_t* samples;    // [S][iters][5]
  int32_t* counts;     // [S][iters]   (-1: the solver failed on that set)
  double* poses;       // [S][iters][12]
  int32_t* idx;        // [n_capacity] inliers of the winner, slot-relative, in order
  int iters;
};

__device__ __forceinline__ int64_t slot_size(const int64_t* slot_base, int s, int64_t cap,
                                             int min_points, int64_t* base, int64_t* rows) {
  *base = slot_base[s];
  *rows = slot_base[s + 1] <= cap ? slot_base[s + 1] - slot_base[s] : 0;   // overflowed: empty
  return *rows >= min_points ? *rows : 0;
}

__global__ __launch_bounds__(64) void cvr_samples(const int64_t* slot_base, int S, int64_t cap,
                                                  CvrWork w) {
  const int s = blockIdx.x;
  if (threadIdx.x != 0) return;
  int64_t base, rows;
  const int64_t n = slot_size(slot_base, s, cap, w.min_points, &base, &rows);
  int32_t* out = w.samples + static_cast<int64_t>(s) * w.iters * EP_SET;
  if (n < EP_SET) return;
  if (n == EP_SET) {                       // RANSACPointSetRegistrator::run: one kernel call
    for (int i = 0; i < EP_SET; ++i) out[i] = i;
    return;
  }
  uint64_t st = 0xffffffffffffffffull;
  const uint32_t nn = static_cast<uint32_t>(n);
  // x % nn without a division in the loop (Granlund-Montgomery, exact for any 32-bit x)
  uint32_t l = 0;
  while ((1ull << l) < nn) ++l;
  const uint32_t mul = static_cast<uint32_t>(((1ull << 32) * ((1ull << l) - nn)) / nn + 1);
  const uint32_t sh1 = l > 0 ? 1 : 0, sh2 = l > 0 ? l - 1 : 0;
  auto draw = [&]() {
    st = static_cast<uint64_t>(static_cast<uint32_t>(st)) * 4164903690u + static_cast<uint32_t>(st >> 32);
    const uint32_t x = static_cast<uint32_t>(st);
    const uint32_t th = __umulhi(mul, x);
    const uint32_t quo = (th + ((x - th) >> sh1)) >> sh2;
    return static_cast<int32_t>(x - quo * nn);
  };
  for (int it = 0; it < w.iters; ++it) {       // the set stays in registers (static indices)
    int32_t s0, s1, s2, s3, s4;
    s0 = draw();
    do { s1 = draw(); } while (s1 == s0);
    do { s2 = draw(); } while (s2 == s0 || s2 == s1);
    do { s3 = draw(); } while (s3 == s0 || s3 == s1 || s3 == s2);
    do { s4 = draw(); } while (s4 == s0 || s4 == s1 || s4 == s2 || s4 == s3);
    int32_t* o = out + it * EP_SET;
    o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = s4;
  }
}

__global__ __launch_bounds__(256) void cvr_hypotheses(
    const double* xy_all, const double* xyz_all, const int64_t* slot_base, int S, int64_t cap,
    const double* Ks, float t2, CvrWork w) {
  __shared__ double s_A[4][144], s_V[4][144], s_R[4][24];
  const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int it = blockIdx.x * 4 + wave;
  int64_t base, rows;
  const int64_t n = slot_size(slot_base, s, cap, w.min_points, &base, &rows);
  if (it >= w.iters || n < EP_SET || (n == EP_SET && it > 0)) return;
  const double* xy = xy_all + 2 * base;
  const double* xyz = xyz_all + 3 * base;
  const double* K = Ks + 9 * s;
  const EpCam cam = {K[0], K[4], K[2], K[5]};
  const int32_t* set = w.samples + (static_cast<int64_t>(s) * w.iters + it) * EP_SET;
  double X[EP_SET][3], x[EP_SET][2];
#pragma unroll
  for (int i = 0; i < EP_SET; ++i) {
    const int64_t p = set[i];
    X[i][0] = f32r(xyz[3 * p]); X[i][1] = f32r(xyz[3 * p + 1]); X[i][2] = f32r(xyz[3 * p + 2]);
    x[i][0] = ep_us_of(f32r(xy[2 * p]), cam.uc, cam.fu);
    x[i][1] = ep_us_of(f32r(xy[2 * p + 1]), cam.vc, cam.fv);
  }
  // sums over the 5 correspondences: plain left to right
  EpFrame f;
  double s3[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < EP_SET; ++i) { s3[0] += X[i][0]; s3[1] += X[i][1]; s3[2] += X[i][2]; }
#pragma unroll
  for (int j = 0; j < 3; ++j) f.c0[j] = s3[j] / static_cast<double>(EP_SET);
  double s6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < EP_SET; ++i) {
    double o[6];
    term_cov(f, X[i], o);
#pragma unroll
    for (int v = 0; v < 6; ++v) s6[v] += o[v];
  }
  int32_t* count_out = w.counts + static_cast<int64_t>(s) * w.iters + it;
  double* pose_out = w.poses + (static_cast<int64_t>(s) * w.iters + it) * 12;
  if (ep_control_points(s6, static_cast<double>(EP_SET), f)) {
    if (lane == 0) *count_out = -1;
    return;
  }
  double s40[40];
#pragma unroll
  for (int v = 0; v < 40; ++v) s40[v] = 0.0;
#pragma unroll
  for (int i = 0; i < EP_SET; ++i) {
    double o[40];
    term_mtm(f, cam, X[i], x[i], o);
#pragma unroll
    for (int v = 0; v < 40; ++v) s40[v] += o[v];
  }
  double* A = s_A[wave];
  double* V = s_V[wave];
  if (lane == 0) ep_write_mtm(s40, cam, A);
  wave_sync();
  jacobi12_wave(A, V, s_R[wave], lane);
  double poses[36];
  bool ok[3];
  ep_candidates(A, V, f, X[0], poses, ok);
  double rep[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < EP_SET; ++i) {
    double o[3];
    term_rep(poses, cam, X[i], x[i], o);
#pragma unroll
    for (int q = 0; q < 3; ++q) rep[q] += o[q];
  }
  const int best = ep_choose(ok, rep);
  if (best < 0) {
    if (lane == 0) *count_out = -1;
    return;
  }
  double P[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) P[i] = best == 0 ? poses[i] : (best == 1 ? poses[12 + i] : poses[24 + i]);
  int cnt = 0;
  if (n == EP_SET) cnt = lane == 0 ? EP_SET : 0;       // the single kernel call: all inliers
  else
    for (int64_t p = lane; p < n; p += 64) cnt += ep_is_inlier(P, cam, xy, xyz, p, t2) ? 1 : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if (lane == 0) {
    *count_out = cnt;
#pragma unroll
    for (int i = 0; i < 12; ++i) pose_out[i] = P[i];
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}

// sums of NV values over the inlier list: 256 strided partials, butterfly, (w0+w1)+(w2+w3)
template <int NV, typename F>
__device__ void block_sums(F term, int64_t m, int t, double* s_red /*[4][NV]*/, double* out) {
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  for (int64_t i = t; i < m; i += 256) {
    double o[NV];
    term(i, o);
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] += o[v];
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const double ws = wave_sum(acc[v]);
    if ((t & 63) == 0) s_red[(t >> 6) * NV + v] = ws;
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v)
    out[v] = (s_red[v] + s_red[NV + v]) + (s_red[2 * NV + v] + s_red[3 * NV + v]);
  __syncthreads();
}

__global__ __launch_bounds__(256) void cvr_select_fit(
    const double* xy_all, const double* xyz_all, const int64_t* slot_base, int S, int64_t cap,
    const double* Ks, float t2, double confidence, CvrWork w, double* poses_out,
    int32_t* success_out, uint8_t* mask_out, int32_t* info_out) {
  __shared__ double s_A[144], s_V[144], s_R[24], s_red[4 * 40], s_pose[12];
  __shared__ int s_best[4], s_wcount[4], s_total;
  const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int64_t base, rows;
  const int64_t n = slot_size(slot_base, s, cap, w.min_points, &base, &rows);
  const double* xy = xy_all + 2 * base;
  const double* xyz = xyz_all + 3 * base;
  uint8_t* mask = mask_out + base;
  const double* K = Ks + 9 * s;
  const EpCam cam = {K[0], K[4], K[2], K[5]};
  const int32_t* counts = w.counts + static_cast<int64_t>(s) * w.iters;
  // replay of the RANSAC loop over the table of inlier counts
  if (t == 0) {
    int niters = w.iters, best_count = 0, best_it = -1, it = 0;
    if (n < EP_SET) niters = 0;
    else if (n == EP_SET) {
      if (counts[0] == EP_SET) { best_it = 0; best_count = EP_SET; }
      niters = 0;
    }
    for (; it < niters; ++it) {
      const int c = counts[it];
      if (c > (best_count > EP_SET - 1 ? best_count : EP_SET - 1)) {
        best_count = c; best_it = it;
        niters = ep_update_niters(confidence, static_cast<double>(n - c) / static_cast<double>(n), niters);
      }
    }
    s_best[0] = best_it; s_best[1] = best_count; s_best[2] = n < EP_SET ? w.iters : niters;
    s_best[3] = it;
  }
  __syncthreads();
  const int best_it = s_best[0];
  if (t < 4 && info_out) info_out[4 * s + t] = s_best[t];
  if (best_it < 0) {
    for (int64_t p = t; p < rows; p += 256) mask[p] = 0;
    if (t == 0) success_out[s] = 0;
    return;
  }
  if (t < 12) s_pose[t] = w.poses[(static_cast<int64_t>(s) * w.iters + best_it) * 12 + t];
  if (t == 0) s_total = 0;
  __syncthreads();
  double P[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) P[i] = s_pose[i];
  // inliers of the winner, compacted in index order
  int32_t* idx = w.idx + base;
  for (int64_t p0 = 0; p0 < n; p0 += 256) {
    const int64_t p = p0 + t;
    const bool in = p < n && (n == EP_SET || ep_is_inlier(P, cam, xy, xyz, p, t2));
    const unsigned long long b = __ballot(in);
    if (lane == 0) s_wcount[wave] = __popcll(b);
    __syncthreads();
    int off = s_total;
    for (int k = 0; k < wave; ++k) off += s_wcount[k];
    if (in) idx[off + __popcll(b & ((1ull << lane) - 1ull))] = static_cast<int32_t>(p);
    if (p < n) mask[p] = in ? 1 : 0;
    __syncthreads();
    if (t == 0) s_total += s_wcount[0] + s_wcount[1] + s_wcount[2] + s_wcount[3];
    __syncthreads();
  }
  const int64_t m = s_total;
  __threadfence_block();
  __syncthreads();
  // EPnP over all inliers
  EpFrame f;
  double s3[3], s6[6], s40[40];
  auto load = [&](int64_t i, double* X, double* x2) {
    const int64_t p = idx[i];
    X[0] = f32r(xyz[3 * p]); X[1] = f32r(xyz[3 * p + 1]); X[2] = f32r(xyz[3 * p + 2]);
    x2[0] = ep_us_of(f32r(xy[2 * p]), cam.uc, cam.fu);
    x2[1] = ep_us_of(f32r(xy[2 * p + 1]), cam.vc, cam.fv);
  };
  block_sums<3>([&](int64_t i, double* o) { double x2[2]; load(i, o, x2); }, m, t, s_red, s3);
#pragma unroll
  for (int j = 0; j < 3; ++j) f.c0[j] = s3[j] / static_cast<double>(m);
  block_sums<6>([&](int64_t i, double* o) { double X[3], x2[2]; load(i, X, x2); term_cov(f, X, o); },
                m, t, s_red, s6);
  bool fail = ep_control_points(s6, static_cast<double>(m), f);
  bool ok[3] = {false, false, false};
  double poses[36];
  int best = -1;
  if (!fail) {                                         // workgroup-uniform
    block_sums<40>([&](int64_t i, double* o) { double X[3], x2[2]; load(i, X, x2); term_mtm(f, cam, X, x2, o); },
                   m, t, s_red, s40);
    if (t == 0) ep_write_mtm(s40, cam, s_A);
    __syncthreads();
    if (wave == 0) jacobi12_wave(s_A, s_V, s_R, lane);
    __syncthreads();
    double Xf[3], xf[2];
    load(0, Xf, xf);
    ep_candidates(s_A, s_V, f, Xf, poses, ok);
    double rep[3];
    block_sums<3>([&](int64_t i, double* o) { double X[3], x2[2]; load(i, X, x2); term_rep(poses, cam, X, x2, o); },
                  m, t, s_red, rep);
    best = ep_choose(ok, rep);
  }
  if (best < 0) {
    for (int64_t p = t; p < n; p += 256) mask[p] = 0;
    if (t == 0) success_out[s] = 0;
    return;
  }
  if (t == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i)
      poses_out[12 * s + i] = best == 0 ? poses[i] : (best == 1 ? poses[12 + i] : poses[24 + i]);
    success_out[s] = 1;
  }
}

struct CvrLayout {
  int64_t samples, counts, poses, idx, total;
};
CvrLayout cvr_layout(int S, int64_t cap, int iters) {
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  CvrLayout l;
  int64_t o = 0;
  l.samples = o; o += up(static_cast<int64_t>(S) * iters * EP_SET * 4);
  l.counts = o; o += up(static_cast<int64_t>(S) * iters * 4);
  l.poses = o; o += up(static_cast<int64_t>(S) * iters * 96);
  l.idx = o; o += up((cap > 0 ? cap : 1) * 4);
  l.total = o;
  return l;
}

int cvr_enqueue(const double* xy, const double* xyz, const int64_t* slot_base, int S,
                int64_t cap, const double* Ks, const EposPnpRansacParams* p, void* work,
                double* poses, int32_t* success, uint8_t* mask, int32_t* info,
                hipStream_t st) {
  const CvrLayout l = cvr_layout(S, cap, p->iterations_count);
  char* wb = static_cast<char*>(work);
  CvrWork w;
  w.samples = reinterpret_cast<int32_t*>(wb + l.samples);
  w.counts = reinterpret_cast<int32_t*>(wb + l.counts);
  w.poses = reinterpret_cast<double*>(wb + l.poses);
  w.idx = reinterpret_cast<int32_t*>(wb + l.idx);
  w.iters = p->iterations_count;
  w.min_points = p->min_point_number;
  const float t2 = static_cast<float>(p->reprojection_error * p->reprojection_error);
  int rc = check_hip(hipMemsetAsync(w.counts, 0xff, static_cast<size_t>(S) * w.iters * 4, st), "memset");
  if (rc) return rc;
  hipLaunchKernelGGL(cvr_samples, dim3(S), dim3(64), 0, st, slot_base, S, cap, w);
  hipLaunchKernelGGL(cvr_hypotheses, dim3((w.iters + 3) / 4, S), dim3(256), 0, st, xy, xyz,
                     slot_base, S, cap, Ks, t2, w);
  hipLaunchKernelGGL(cvr_select_fit, dim3(S), dim3(256), 0, st, xy, xyz, slot_base, S, cap, Ks,
                     t2, p->confidence, w, poses, success, mask, info);
  return launch_status("cvr_select_fit");
}

}  // namespace
}  // namespace epos

using namespace epos;

extern "C" void epos_pnp_ransac_params_default(EposPnpRansacParams* p) {
  if (!p) return;
  p->iterations_count = 400;        // max_fitting_iterations, scripts/infer.py:87-89
  p->reprojection_error = 4.0;      // inlier_thresh, scripts/infer.py:76-78
  p->confidence = 0.99;             // scripts/infer.py:517
  p->min_point_number = 0;          // OpenCV's own rule; the script skips n < 6 (infer.py:420)
}

extern "C" int64_t epos_pnp_ransac_workspace_bytes(int S, int64_t n_capacity,
                                                   const EposPnpRansacParams* p) {
  if (!p || S < 0 || n_capacity < 0 || p->iterations_count < 1) return EPOS_E_INVALID;
  return cvr_layout(S, n_capacity, p->iterations_count).total;
}

extern "C" int epos_solve_pnp_ransac_device(
    const double* xy, const double* xyz, const int64_t* slot_base, int S, int64_t n_capacity,
    const double* Ks, const EposPnpRansacParams* p, void* work, double* poses,
    int32_t* success, uint8_t* inlier_mask, int32_t* info, void* stream) {
  EPOS_REQUIRE(xy && xyz && slot_base && Ks && p && work && poses && success && inlier_mask,
               "null pointer");
  EPOS_REQUIRE(p->iterations_count >= 1 && p->iterations_count <= 1000000,
               "iterations_count must be in [1, 1e6]");
  EPOS_REQUIRE(p->reprojection_error > 0.0, "reprojection_error must be positive");
  if (S == 0) return EPOS_OK;
  return cvr_enqueue(xy, xyz, slot_base, S, n_capacity, Ks, p, work, poses, success,
                     inlier_mask, info, static_cast<hipStream_t>(stream));
}

extern "C" int epos_solve_pnp_ransac(const double* xy, const double* xyz, int64_t n,
                                     const double* K, const EposPnpRansacParams* p,
                                     double* pose_out, uint8_t* inlier_mask_out,
                                     int32_t* info_out) {
  EPOS_REQUIRE(K && p && pose_out && (n == 0 || (xy && xyz && inlier_mask_out)), "null pointer");
  EPOS_REQUIRE(p->iterations_count >= 1 && p->iterations_count <= 1000000,
               "iterations_count must be in [1, 1e6]");
  EPOS_REQUIRE(p->reprojection_error > 0.0, "reprojection_error must be positive");
  if (info_out) { info_out[0] = -1; info_out[1] = 0; info_out[2] = p->iterations_count; info_out[3] = 0; }
  for (int64_t i = 0; i < n; ++i) inlier_mask_out[i] = 0;
  if (n < EP_SET || n < p->min_point_number) return 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    set_error("epos_solve_pnp_ransac: no HIP device (there is no CPU fallback)");
    return EPOS_E_NODEVICE;
  }
  const int64_t wbytes = epos_pnp_ransac_workspace_bytes(1, n, p);
  const int64_t sizes[] = {n * 16, n * 24, 16, 72, wbytes, 96, 8, n, 16};
  constexpr int ND = 9;
  char* d[ND] = {0};
  int rc = EPOS_OK;
  for (int i = 0; i < ND && !rc; ++i)
    rc = check_hip(hipMalloc(reinterpret_cast<void**>(&d[i]), sizes[i] + 8), "hipMalloc");
  int32_t ok = 0;
  if (!rc) {
    const int64_t sb[2] = {0, n};
    rc = check_hip(hipMemcpy(d[0], xy, n * 16, hipMemcpyHostToDevice), "copy xy");
    if (!rc) rc = check_hip(hipMemcpy(d[1], xyz, n * 24, hipMemcpyHostToDevice), "copy xyz");
    if (!rc) rc = check_hip(hipMemcpy(d[2], sb, 16, hipMemcpyHostToDevice), "copy base");
    if (!rc) rc = check_hip(hipMemcpy(d[3], K, 72, hipMemcpyHostToDevice), "copy K");
    if (!rc)
      rc = cvr_enqueue(reinterpret_cast<double*>(d[0]), reinterpret_cast<double*>(d[1]),
                       reinterpret_cast<int64_t*>(d[2]), 1, n, reinterpret_cast<double*>(d[3]),
                       p, d[4], reinterpret_cast<double*>(d[5]), reinterpret_cast<int32_t*>(d[6]),
                       reinterpret_cast<uint8_t*>(d[7]), reinterpret_cast<int32_t*>(d[8]), nullptr);
    if (!rc) rc = check_hip(hipDeviceSynchronize(), "sync");
    if (!rc) rc = check_hip(hipMemcpy(&ok, d[6], 4, hipMemcpyDeviceToHost), "copy success");
    if (!rc && ok) rc = check_hip(hipMemcpy(pose_out, d[5], 96, hipMemcpyDeviceToHost), "copy pose");
    if (!rc) rc = check_hip(hipMemcpy(inlier_mask_out, d[7], n, hipMemcpyDeviceToHost), "copy mask");
    if (!rc && info_out) rc = check_hip(hipMemcpy(info_out, d[8], 16, hipMemcpyDeviceToHost), "copy info");
  }
  for (int i = 0; i < ND; ++i)
    if (d[i]) (void)hipFree(d[i]);
  return rc ? rc : ok;
}
