// Two-view linear triangulation on the GPU: epipolar.triangulate_matlab (reference
// multiviewunsynch/reconstruction/epipolar.py:497-510) for N point pairs, one lane per pair, plus the two
// reprojection errors Scene.triangulate filters by (common.py:786-789, epipolar.py:639).
//
// The reference takes, per point, the right singular vector of the smallest singular value of the 4x4 matrix
//     A = [ u1 P1[2] - P1[0] ;  v1 P1[2] - P1[1] ;  u2 P2[2] - P2[0] ;  v2 P2[2] - P2[1] ]
// (np.linalg.svd, V[-1] / V[-1,-1]).  Here: one-sided Jacobi (Hestenes) on the columns of A -- plane rotations applied to
// A and accumulated in V until all column pairs are orthogonal; the column of smallest norm is that singular vector.
// It works on A itself (not on A^T A), so small singular values keep their relative accuracy like LAPACK's SVD does; the
// scale/sign ambiguity disappears in the division by the last component, as in the reference.
#pragma once
#include "ba_math.h"      // MVUS_HD; pulls in hip_runtime.h under hipcc (the test-only host harness compiles the pair math with g++)

namespace mvus {

MVUS_HD void triangulate_pair(const double P1[12], const double P2[12], double u1, double v1, double u2, double v2, double X[4]) {
  double G[4][4], V[4][4];                       // G[row][col]
  for (int j = 0; j < 4; ++j) {
    G[0][j] = u1 * P1[8 + j] - P1[j];
    G[1][j] = v1 * P1[8 + j] - P1[4 + j];
    G[2][j] = u2 * P2[8 + j] - P2[j];
    G[3][j] = v2 * P2[8 + j] - P2[4 + j];
    for (int i = 0; i < 4; ++i) V[i][j] = i == j ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        double al = 0.0, be = 0.0, ga = 0.0;
        for (int i = 0; i < 4; ++i) { al += G[i][p] * G[i][p]; be += G[i][q] * G[i][q]; ga += G[i][p] * G[i][q]; }
        if (fabs(ga) <= 1e-16 * sqrt(al * be) || ga == 0.0) continue;
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 4; ++i) {
          const double gp = G[i][p], gq = G[i][q];
          G[i][p] = c * gp - s * gq; G[i][q] = s * gp + c * gq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  int jm = 0;
  double nm = 0.0;
  for (int j = 0; j < 4; ++j) {
    double nj = 0.0;
    for (int i = 0; i < 4; ++i) nj += G[i][j] * G[i][j];
    if (j == 0 || nj < nm) { nm = nj; jm = j; }
  }
  // static indexing of V's columns (the column index is data dependent): select
  for (int i = 0; i < 4; ++i) X[i] = jm == 0 ? V[i][0] : (jm == 1 ? V[i][1] : (jm == 2 ? V[i][2] : V[i][3]));
  const double w = X[3];
  for (int i = 0; i < 4; ++i) X[i] = X[i] / w;
}

// Camera.projectPoint (common.py:1072-1079) + epipolar.reprojection_error (epipolar.py:639)
MVUS_HD double reprojection_distance(const double P[12], const double X[4], double u, double v) {
  const double a = P[0] * X[0] + P[1] * X[1] + P[2] * X[2] + P[3] * X[3];
  const double b = P[4] * X[0] + P[5] * X[1] + P[6] * X[2] + P[7] * X[3];
  const double c = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11] * X[3];
  const double du = u - a / c, dv = v - b / c;
  return sqrt(du * du + dv * dv);
}

#if defined(__HIPCC__)
struct TriCams { double P1[12], P2[12]; };      // by value in the kernel arguments: wave-uniform -> SGPRs

// x1, x2: [2][N] (u row, v row); X: [4][N]; err1/err2: [N] or nullptr.  Every access is one contiguous 512-B segment per wavefront.
__global__ __launch_bounds__(256) void k_triangulate(TriCams cams, long long N, const double* __restrict__ x1, const double* __restrict__ x2,
                                                     double* __restrict__ X, double* __restrict__ err1, double* __restrict__ err2) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= N) return;
  const double u1 = x1[i], v1 = x1[N + i], u2 = x2[i], v2 = x2[N + i];
  double Xh[4];
  triangulate_pair(cams.P1, cams.P2, u1, v1, u2, v2, Xh);
  X[i] = Xh[0]; X[N + i] = Xh[1]; X[2 * N + i] = Xh[2]; X[3 * N + i] = Xh[3];
  if (err1) err1[i] = reprojection_distance(cams.P1, Xh, u1, v1);
  if (err2) err2[i] = reprojection_distance(cams.P2, Xh, u2, v2);
}
#endif

}  // namespace mvus
