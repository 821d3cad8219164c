// Spline maintenance either side of BA, on the GPU:
//   k_spline_eval     Scene.spline_to_traj (reference common.py:273-301): X(t) of the interval each timestamp belongs to
//                     (closed ends, common.py:292) -- scipy splev == FITPACK splev.f/fpbspl.f, the same recurrence
//                     (bspline_basis) the BA kernels use
//   k_lsq_*           least-squares coefficients of a cubic spline on a FIXED knot vector for data (t_i, X_i): the banded
//                     normal equations B^T B c = B^T X (bandwidth 4) accumulated by one lane per data point, then a banded
//                     Cholesky solve by one wavefront.  The reference's traj_to_spline lets FITPACK choose the knots
//                     adaptively inside its smooth_factor loop (common.py:224-270); that search stays on the host
//                     (Scene.traj_to_spline), this is the refit on the knots it found -- scipy's make_lsq_spline is the oracle.
#pragma once
#include "ba_math.h"

namespace mvus {

#if defined(__HIPCC__)
struct SplineSet {               // S splines as they sit in device memory
  int S;
  const double* istart;          // [S]
  const double* iend;            // [S]
  const long long* knot_off;     // [S+1]
  const double* knots;
  const long long* coef_off;     // [S+1] offset of spline s' coefficient block (cx(n) cy(n) cz(n)) in coefs, in doubles
  const double* coefs;
};

__global__ __launch_bounds__(256) void k_spline_eval(SplineSet sp, long long nt, const double* __restrict__ t, double* __restrict__ X,
                                                     int32_t* __restrict__ which) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= nt) return;
  const double x = t[i];
  int lo = 0, hi = sp.S;                              // last interval whose start <= x
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (x >= sp.istart[mid]) lo = mid; else hi = mid; }
  const bool in = x >= sp.istart[lo] && x <= sp.iend[lo];        // closed at both ends (common.py:292)
  which[i] = in ? lo : -1;
  double o0 = 0.0, o1 = 0.0, o2 = 0.0;
  if (in) {
    const double* k = sp.knots + sp.knot_off[lo];
    const int n = (int)(sp.knot_off[lo + 1] - sp.knot_off[lo]) - 4;
    const int l = find_span(k, n, x);
    double h[4], dh[4];
    bspline_basis<false>(k, l, x, h, dh);
    const double* c = sp.coefs + sp.coef_off[lo] + (l - 3);
#pragma unroll
    for (int q = 0; q < 4; ++q) { o0 = o0 + c[q] * h[q]; o1 = o1 + c[n + q] * h[q]; o2 = o2 + c[2 * n + q] * h[q]; }
  }
  X[i] = o0; X[nt + i] = o1; X[2 * nt + i] = o2;
}

// ---- least squares on fixed knots ------------------------------------------------------------------------------------
// Normal equations in lower banded storage: G[j][w] = (B^T B)(j, j - w), w = 0..3;  rhs[d][j] = (B^T X_d)(j).
__global__ __launch_bounds__(256) void k_lsq_accumulate(const double* __restrict__ knots, int n, long long m, const double* __restrict__ t,
                                                        const double* __restrict__ X, double* __restrict__ G, double* __restrict__ rhs) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= m) return;
  const double x = t[i];
  const int l = find_span(knots, n, x);
  double h[4], dh[4];
  bspline_basis<false>(knots, l, x, h, dh);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int ja = l - 3 + a;
#pragma unroll
    for (int b = 0; b <= a; ++b) unsafeAtomicAdd(&G[4 * ja + (a - b)], h[a] * h[b]);
#pragma unroll
    for (int d = 0; d < 3; ++d) unsafeAtomicAdd(&rhs[(long long)d * n + ja], h[a] * X[(long long)d * m + i]);
  }
}

// banded Cholesky G = L L^T (L keeps the band) and the three solves, one wavefront: lanes 0..2 carry the three right-hand
// sides through the substitutions, the factorisation itself is a 4-wide recurrence done redundantly by every lane
__global__ __launch_bounds__(64) void k_lsq_solve(int n, double* __restrict__ G, double* __restrict__ rhs, int* __restrict__ fail) {
  const int lane = threadIdx.x;
  // factorise in place: L(j, j-w) in G[4j + w]
  for (int j = 0; j < n; ++j) {
    double row[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) row[w] = G[4 * j + w];
    // L(j, j-w) for w = 3, 2, 1, then the diagonal
#pragma unroll
    for (int w = 3; w >= 1; --w) {
      const int k = j - w;
      if (k < 0) { row[w] = 0.0; continue; }
      double v = row[w];
      // subtract sum over p < k of L(j,p) L(k,p), p >= j-3
#pragma unroll
      for (int u = w + 1; u <= 3; ++u) {           // p = j - u
        const int p = j - u;
        if (p >= 0) v -= row[u] * G[4 * k + (u - w)];
      }
      row[w] = v / G[4 * k];
    }
    double d = row[0];
#pragma unroll
    for (int u = 1; u <= 3; ++u) if (j - u >= 0) d -= row[u] * row[u];
    if (!(d > 0.0)) { if (lane == 0) fail[0] = 1; d = 1.0; }
    row[0] = sqrt(d);
    __syncthreads();                                 // every lane computed the same row; one writes it
    if (lane == 0) {
#pragma unroll
      for (int w = 0; w < 4; ++w) G[4 * j + w] = row[w];
    }
    __syncthreads();
  }
  if (lane >= 3) return;
  double* b = rhs + (long long)lane * n;
  for (int j = 0; j < n; ++j) {                      // L y = b
    double v = b[j];
#pragma unroll
    for (int u = 1; u <= 3; ++u) if (j - u >= 0) v -= G[4 * j + u] * b[j - u];
    b[j] = v / G[4 * j];
  }
  for (int j = n - 1; j >= 0; --j) {                 // L^T c = y
    double v = b[j];
#pragma unroll
    for (int u = 1; u <= 3; ++u) if (j + u < n) v -= G[4 * (j + u) + u] * b[j + u];
    b[j] = v / G[4 * j];
  }
}
#endif

}  // namespace mvus
