// Absolute pose of a new camera from 3-D points on the trajectory and their detections: Scene.get_camera_pose (reference
// common.py:719-750), which hands the problem to cv2.solvePnPRansac(objectPoints, imagePoints, K, d, reprojectionError=error)
// -- OpenCV, a third-party dependency that is absent from this image, so there is no reference output to pin against
// ("parity unpinned").  What is restated is the published contract of that call with its defaults (iterationsCount
// = 100, confidence = 0.99, SOLVEPNP_ITERATIVE): RANSAC over minimal-sample poses scored by the number of points whose
// reprojection error (pixels, with the 5-coefficient distortion model) is below the threshold, then an iterative
// least-squares refinement of the best pose on its inliers.  The mapping onto the GPU is this build's own:
//   k_pnp_hypotheses   one lane per hypothesis: six sampled points -> direct linear transform (smallest eigenvector of the
//                      12x12 normal matrix by inverse iteration on its Cholesky factor) -> nearest rotation (polar iteration)
//                      and translation.  OpenCV draws 5 points for EPnP one hypothesis after the other and stops early by its
//                      confidence rule; here all hypotheses are evaluated at once, so the early stop has nothing to save.
//   k_pnp_score        one workgroup per hypothesis: inlier count over all points
//   k_pnp_mask         inlier flags of the winning hypothesis
//   k_pnp_normal       Gauss-Newton normal equations (6x6) of the pixel reprojection error over the inliers, pose perturbed
//                      on the left (R <- exp([w]x) R); the damped iteration itself runs on the host (mvus_pnp_ransac)
// The final pose is the minimiser of the reprojection error over the inliers -- the quantity OpenCV's refinement also
// minimises -- and is checked against an independent minimiser (oracle/pnp_oracle.py) and against ground truth.
#pragma once
#include <cmath>
#include <cstdint>

#include "ba_math.h"

namespace mvus {

// forward 5-coefficient distortion (k1 k2 p1 p2 k3) of normalised coordinates; Jd = d(xd, yd)/d(x, y) row-major when asked for
MVUS_HD void distort5(double x, double y, const double* d, double& xd, double& yd, double* Jd) {
  const double r2 = x * x + y * y;
  const double rad = 1.0 + r2 * (d[0] + r2 * (d[1] + r2 * d[4]));
  const double drad = d[0] + r2 * (2.0 * d[1] + 3.0 * r2 * d[4]);           // d rad / d r2
  xd = x * rad + 2.0 * d[2] * x * y + d[3] * (r2 + 2.0 * x * x);
  yd = y * rad + d[2] * (r2 + 2.0 * y * y) + 2.0 * d[3] * x * y;
  if (Jd) {
    Jd[0] = rad + 2.0 * x * x * drad + 2.0 * d[2] * y + 6.0 * d[3] * x;
    Jd[1] = 2.0 * x * y * drad + 2.0 * d[2] * x + 2.0 * d[3] * y;
    Jd[2] = 2.0 * x * y * drad + 2.0 * d[2] * x + 2.0 * d[3] * y;
    Jd[3] = rad + 2.0 * y * y * drad + 6.0 * d[2] * y + 2.0 * d[3] * x;
  }
}

// pixel of X under pose (R row-major, t), intrinsics K = fx fy cx cy and distortion d; false when the point is not in front
MVUS_HD bool pnp_project(const double* K, const double* d, const double* R, const double* t, const double* X, double& u, double& v) {
  const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
  const double yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
  const double zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
  if (!(zc > 0.0)) return false;
  double xd, yd;
  distort5(xc / zc, yc / zc, d, xd, yd, nullptr);
  u = K[0] * xd + K[2];
  v = K[1] * yd + K[3];
  return true;
}

MVUS_HD double det3(const double* A) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}
MVUS_HD void inv_transpose3(const double* A, double det, double* out) {       // A^-T = cofactor matrix / det
  out[0] = (A[4] * A[8] - A[5] * A[7]) / det; out[1] = (A[5] * A[6] - A[3] * A[8]) / det; out[2] = (A[3] * A[7] - A[4] * A[6]) / det;
  out[3] = (A[2] * A[7] - A[1] * A[8]) / det; out[4] = (A[0] * A[8] - A[2] * A[6]) / det; out[5] = (A[1] * A[6] - A[0] * A[7]) / det;
  out[6] = (A[1] * A[5] - A[2] * A[4]) / det; out[7] = (A[2] * A[3] - A[0] * A[5]) / det; out[8] = (A[0] * A[4] - A[1] * A[3]) / det;
}

// Pose from six 3-D points Xs[6][3] and their normalised, undistorted image coordinates xn[6][2] by the direct linear
// transform: P (3x4, up to scale) = the eigenvector of the smallest eigenvalue of A^T A (12x12; two rows per point),
// found by inverse iteration; then P = s [R | t] with R the rotation nearest to the left 3x3 block.  false: degenerate sample.
// (The 12x12 matrix is kept as its packed lower triangle, 78 doubles, and every loop over it is unrolled: all indices are compile-time
// constants, so on the GPU the triangle lives in registers -- as a [12][12] array indexed by loop variables it was 1 216 bytes of
// scratch per lane, rounds 2 - 4.)
#define MVUS_TRI(a, b) ((a) * ((a) + 1) / 2 + (b))
MVUS_HD bool pnp_dlt6(const double (*Xs)[3], const double (*xn)[2], double* R, double* t) {
  double M[78];
#pragma unroll
  for (int e = 0; e < 78; ++e) M[e] = 0.0;
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const double X = Xs[p][0], Y = Xs[p][1], Z = Xs[p][2], x = xn[p][0], y = xn[p][1];
    const double r1[12] = {X, Y, Z, 1.0, 0.0, 0.0, 0.0, 0.0, -x * X, -x * Y, -x * Z, -x};
    const double r2[12] = {0.0, 0.0, 0.0, 0.0, X, Y, Z, 1.0, -y * X, -y * Y, -y * Z, -y};
#pragma unroll
    for (int a = 0; a < 12; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) M[MVUS_TRI(a, b)] += r1[a] * r1[b] + r2[a] * r2[b];
  }
  double tr = 0.0;
#pragma unroll
  for (int a = 0; a < 12; ++a) tr += M[MVUS_TRI(a, a)];
  if (!(tr > 0.0)) return false;
  const double mu = 1e-13 * tr;
#pragma unroll
  for (int a = 0; a < 12; ++a) M[MVUS_TRI(a, a)] += mu;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < 12; ++j) {                                            // Cholesky, lower triangle in place
    double dsum = M[MVUS_TRI(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) dsum -= M[MVUS_TRI(j, k)] * M[MVUS_TRI(j, k)];
    bad |= !(dsum > 0.0);
    const double l = sqrt(dsum > 0.0 ? dsum : 1.0);
    M[MVUS_TRI(j, j)] = l;
#pragma unroll
    for (int i = j + 1; i < 12; ++i) {
      double sacc = M[MVUS_TRI(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) sacc -= M[MVUS_TRI(i, k)] * M[MVUS_TRI(j, k)];
      M[MVUS_TRI(i, j)] = sacc / l;
    }
  }
  if (bad) return false;
  double v[12];
#pragma unroll
  for (int a = 0; a < 12; ++a) v[a] = 1.0 + 0.37 * a - 0.11 * a * a;          // a fixed start that is not an eigenvector of anything in particular
  for (int it = 0; it < 6; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      double sacc = v[i];
#pragma unroll
      for (int k = 0; k < i; ++k) sacc -= M[MVUS_TRI(i, k)] * v[k];
      v[i] = sacc / M[MVUS_TRI(i, i)];
    }
#pragma unroll
    for (int i = 11; i >= 0; --i) {
      double sacc = v[i];
#pragma unroll
      for (int k = i + 1; k < 12; ++k) sacc -= M[MVUS_TRI(k, i)] * v[k];
      v[i] = sacc / M[MVUS_TRI(i, i)];
    }
    double nn = 0.0;
#pragma unroll
    for (int a = 0; a < 12; ++a) nn += v[a] * v[a];
    if (!(nn > 0.0) || !(nn < 1e300)) return false;
    nn = 1.0 / sqrt(nn);
#pragma unroll
    for (int a = 0; a < 12; ++a) v[a] *= nn;
  }
  double A[9] = {v[0], v[1], v[2], v[4], v[5], v[6], v[8], v[9], v[10]};
  double tt[3] = {v[3], v[7], v[11]};
  double dA = det3(A);
  if (dA < 0.0) { for (int a = 0; a < 9; ++a) A[a] = -A[a]; for (int a = 0; a < 3; ++a) tt[a] = -tt[a]; dA = -dA; }
  double fro = 0.0;
  for (int a = 0; a < 9; ++a) fro += A[a] * A[a];
  if (!(dA > 1e-12 * fro * sqrt(fro))) return false;
  const double sc = sqrt(3.0 / fro);
  for (int a = 0; a < 9; ++a) R[a] = A[a] * sc;
  for (int it = 0; it < 20; ++it) {                                          // polar decomposition: R <- (R + R^-T) / 2
    const double dr = det3(R);
    if (!(dr > 1e-9)) return false;
    double Rit[9];
    inv_transpose3(R, dr, Rit);
    double change = 0.0;
    for (int a = 0; a < 9; ++a) { const double nv = 0.5 * (R[a] + Rit[a]); change += (nv - R[a]) * (nv - R[a]); R[a] = nv; }
    if (change < 1e-30) break;
  }
  double lam = 0.0;                                                          // A = lam R in the least-squares sense
  for (int a = 0; a < 9; ++a) lam += R[a] * A[a];
  lam /= 3.0;
  if (!(lam > 0.0)) return false;
  for (int a = 0; a < 3; ++a) t[a] = tt[a] / lam;
  for (int p = 0; p < 6; ++p) {                                              // all six in front of the camera
    const double zc = R[6] * Xs[p][0] + R[7] * Xs[p][1] + R[8] * Xs[p][2] + t[2];
    if (!(zc > 0.0)) return false;
  }
  return true;
}

MVUS_HD unsigned long long pnp_mix(unsigned long long z) {                   // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
// the six distinct sample indices of hypothesis h (counter based: the same on every run and on the host)
MVUS_HD void pnp_sample6(unsigned long long seed, int h, long long N, long long* idx) {
  unsigned long long ctr = seed * 0x100000001b3ull + (unsigned long long)h * 1000003ull;
  for (int k = 0; k < 6;) {
    ctr = pnp_mix(ctr);
    const long long c = (long long)(ctr % (unsigned long long)N);
    bool dup = false;
    for (int j = 0; j < k; ++j) dup |= idx[j] == c;
    if (!dup) idx[k++] = c;
  }
}

// one point's contribution to the Gauss-Newton system of the pixel reprojection error, pose perturbed on the left:
// acc[0..20] lower triangle of J^T J (row-major a >= b), acc[21..26] J^T r, acc[27] |r|^2.  false: behind the camera.
MVUS_HD bool pnp_point_normal(const double* K, const double* d, const double* R, const double* t, const double* X, double u, double v, double* acc) {
  const double rx = R[0] * X[0] + R[1] * X[1] + R[2] * X[2], ry = R[3] * X[0] + R[4] * X[1] + R[5] * X[2], rz = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
  const double xc = rx + t[0], yc = ry + t[1], zc = rz + t[2];
  if (!(zc > 0.0)) return false;
  const double iz = 1.0 / zc, x = xc * iz, y = yc * iz;
  double xd, yd, Jd[4];
  distort5(x, y, d, xd, yd, Jd);
  const double ru = K[0] * xd + K[2] - u, rv = K[1] * yd + K[3] - v;
  // d(x, y)/d(Xc) = [iz 0 -x iz; 0 iz -y iz];  d(Xc)/d(w) = -[R X]x,  d(Xc)/d(dt) = I
  const double px[3] = {iz, 0.0, -x * iz}, py[3] = {0.0, iz, -y * iz};
  double gu[3], gv[3];                                                       // d(u, v)/d(Xc)
  for (int a = 0; a < 3; ++a) { gu[a] = K[0] * (Jd[0] * px[a] + Jd[1] * py[a]); gv[a] = K[1] * (Jd[2] * px[a] + Jd[3] * py[a]); }
  // row vectors g (-[r]x) = r x g ... with (-[r]x) columns: d/dw_j = -(e_j x r) . g = (r x g)_j reversed sign: use cross(r, g)
  const double r3[3] = {rx, ry, rz};
  double ju[6], jv[6];
  ju[0] = r3[1] * gu[2] - r3[2] * gu[1]; ju[1] = r3[2] * gu[0] - r3[0] * gu[2]; ju[2] = r3[0] * gu[1] - r3[1] * gu[0];
  jv[0] = r3[1] * gv[2] - r3[2] * gv[1]; jv[1] = r3[2] * gv[0] - r3[0] * gv[2]; jv[2] = r3[0] * gv[1] - r3[1] * gv[0];
  for (int a = 0; a < 3; ++a) { ju[3 + a] = gu[a]; jv[3 + a] = gv[a]; }
  int e = 0;
  for (int a = 0; a < 6; ++a) for (int b = 0; b <= a; ++b) acc[e++] += ju[a] * ju[b] + jv[a] * jv[b];
  for (int a = 0; a < 6; ++a) acc[21 + a] += ju[a] * ru + jv[a] * rv;
  acc[27] += ru * ru + rv * rv;
  return true;
}

// rotation matrix (row-major) -> rotation vector (cv2.Rodrigues matrix -> vector)
MVUS_HD void rotation_to_rvec(const double* R, double* r) {
  const double cs = fmin(1.0, fmax(-1.0, 0.5 * (R[0] + R[4] + R[8] - 1.0)));
  const double th = acos(cs);
  const double ax = R[7] - R[5], ay = R[2] - R[6], az = R[3] - R[1];        // 2 sin(th) * axis
  const double s2 = sqrt(ax * ax + ay * ay + az * az);
  if (th < 1e-8 || s2 < 1e-300) { r[0] = 0.5 * ax; r[1] = 0.5 * ay; r[2] = 0.5 * az; return; }
  if (s2 > 1e-6) { const double f = th / s2; r[0] = f * ax; r[1] = f * ay; r[2] = f * az; return; }
  // th close to pi: the axis from the diagonal of (R + I) / 2 = axis axis^T, signs from the off-diagonal sums
  double xx = sqrt(fmax(0.0, 0.5 * (R[0] + 1.0))), yy = sqrt(fmax(0.0, 0.5 * (R[4] + 1.0))), zz = sqrt(fmax(0.0, 0.5 * (R[8] + 1.0)));
  if (xx >= yy && xx >= zz) { yy = (R[1] + R[3] >= 0.0) ? yy : -yy; zz = (R[2] + R[6] >= 0.0) ? zz : -zz; }
  else if (yy >= zz) { xx = (R[1] + R[3] >= 0.0) ? xx : -xx; zz = (R[5] + R[7] >= 0.0) ? zz : -zz; }
  else { xx = (R[2] + R[6] >= 0.0) ? xx : -xx; yy = (R[5] + R[7] >= 0.0) ? yy : -yy; }
  if (ax * xx + ay * yy + az * zz < 0.0) { xx = -xx; yy = -yy; zz = -zz; }
  const double nn = sqrt(xx * xx + yy * yy + zz * zz);
  r[0] = th * xx / nn; r[1] = th * yy / nn; r[2] = th * zz / nn;
}

#if defined(__HIPCC__)
// undistorted normalised image coordinates of the raw pixels (the fixed-point iteration of cv2.undistortPoints)
__global__ __launch_bounds__(256) void k_pnp_normalise(long long N, const double* __restrict__ uv, const double* __restrict__ Kd, double* __restrict__ xn) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= N) return;
  double xo, yo;
  undistort5<false>((uv[i] - Kd[2]) / Kd[0], (uv[N + i] - Kd[3]) / Kd[1], Kd + 4, xo, yo, nullptr, nullptr);
  xn[i] = xo; xn[N + i] = yo;
}

// X: x(N) y(N) z(N); xn: undistorted normalised image coordinates x(N) y(N); poses: per hypothesis R(9) t(3) ok(1) = 13 doubles
__global__ __launch_bounds__(64) void k_pnp_hypotheses(int H, unsigned long long seed, long long N, const double* __restrict__ X, const double* __restrict__ xn,
                                                       double* __restrict__ poses) {
  const int h = blockIdx.x * 64 + threadIdx.x;
  if (h >= H) return;
  long long idx[6];
  pnp_sample6(seed, h, N, idx);
  double Xs[6][3], xs[6][2];
  for (int k = 0; k < 6; ++k) {
    for (int a = 0; a < 3; ++a) Xs[k][a] = X[(long long)a * N + idx[k]];
    xs[k][0] = xn[idx[k]]; xs[k][1] = xn[N + idx[k]];
  }
  double R[9], t[3];
  const bool ok = pnp_dlt6(Xs, xs, R, t);
  double* out = poses + 13ll * h;
  for (int a = 0; a < 9; ++a) out[a] = ok ? R[a] : 0.0;
  for (int a = 0; a < 3; ++a) out[9 + a] = ok ? t[a] : 0.0;
  out[12] = ok ? 1.0 : 0.0;
}

__global__ __launch_bounds__(256) void k_pnp_score(long long N, const double* __restrict__ X, const double* __restrict__ uv, const double* __restrict__ Kd,
                                                   const double* __restrict__ poses, double thr2, int32_t* __restrict__ counts) {
  __shared__ int red[256];
  const double* pose = poses + 13ll * blockIdx.x;
  int cnt = 0;
  if (pose[12] != 0.0) {
    for (long long i = threadIdx.x; i < N; i += 256) {
      const double Xi[3] = {X[i], X[N + i], X[2 * N + i]};
      double u, v;
      if (pnp_project(Kd, Kd + 4, pose, pose + 9, Xi, u, v)) {
        const double du = u - uv[i], dv = v - uv[N + i];
        cnt += (du * du + dv * dv <= thr2) ? 1 : 0;
      }
    }
  }
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) counts[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_pnp_mask(long long N, const double* __restrict__ X, const double* __restrict__ uv, const double* __restrict__ Kd,
                                                  const double* __restrict__ pose, double thr2, uint8_t* __restrict__ mask) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= N) return;
  const double Xi[3] = {X[i], X[N + i], X[2 * N + i]};
  double u, v;
  bool in = false;
  if (pnp_project(Kd, Kd + 4, pose, pose + 9, Xi, u, v)) { const double du = u - uv[i], dv = v - uv[N + i]; in = du * du + dv * dv <= thr2; }
  mask[i] = in ? 1 : 0;
}

// out[0..27] (pnp_point_normal's layout) summed over the flagged points, out[28] = points behind the camera; one workgroup
__global__ __launch_bounds__(256) void k_pnp_normal(long long N, const double* __restrict__ X, const double* __restrict__ uv, const double* __restrict__ Kd,
                                                    const double* __restrict__ pose, const uint8_t* __restrict__ mask, double* __restrict__ out) {
  __shared__ double red[29][256];
  double acc[29];
  for (int k = 0; k < 29; ++k) acc[k] = 0.0;
  for (long long i = threadIdx.x; i < N; i += 256) {
    if (!mask[i]) continue;
    const double Xi[3] = {X[i], X[N + i], X[2 * N + i]};
    if (!pnp_point_normal(Kd, Kd + 4, pose, pose + 9, Xi, uv[i], uv[N + i], acc)) acc[28] += 1.0;
  }
  for (int k = 0; k < 29; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) for (int k = 0; k < 29; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x < 29) out[threadIdx.x] = red[threadIdx.x][0];
}
#endif

}  // namespace mvus
