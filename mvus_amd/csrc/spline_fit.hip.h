// Smoothing-spline fit of a 3-D trajectory on the GPU: Scene.traj_to_spline's scipy.interpolate.splprep(X, u=t, s=s, k=3)
// (reference common.py:247, 267), i.e. FITPACK parcur / fppara (Dierckx) with iopt = 0, w = 1, nest = m + 2k.
//
// Same algorithm, different linear algebra.  fppara's control flow runs on the host (fit_smoothing_spline below: knots are
// added where the residual of the least-squares spline is largest until f(p = inf) <= s, then the smoothing parameter p with
// F(p) = s is found by rational interpolation -- fpknot, fprati and fpdisc are restated line by line, they are O(knots) integer
// / scalar work).  Everything that touches the m samples or solves a system runs on the device:
//   k_fit_basis      knot span + the four cubic B-splines of every sample (the fpbspl recurrence of ba_math.h)
//   k_fit_blocks     per knot span: sum of h h^T and h x^T over its samples (fixed-order tree per workgroup)
//   k_fit_band       banded normal equations A^T A (half-bandwidth 3) and A^T X from <= 4 span blocks per entry
//   k_fit_penalty    B^T B of fpdisc's discontinuity-jump matrix (half-bandwidth 4);  k_fit_combine: A^T A + B^T B / p^2 (fppara rotates the rows of B in with weight 1/p)
//   k_band_solve     banded Cholesky + both substitutions, one wavefront, the previous rows in registers; in fp64, and again in
//                    double-double when a pivot is lost to rounding (knot sets close to interpolation: cond(A) ~ 1e10)
//   k_fit_residual   per-sample squared residual;  k_fit_fpint: per-span sums with FITPACK's half/half rule for the sample
//                    that sits on a knot, and the total f_p
// FITPACK rotates every observation row (and, in the p iteration, every row of B all the way down the band: O(n^2)) into a
// triangle with Givens rotations, one after the other; the normal equations give the same splines (same knots on every fixture
// and seeded case of tests/test_traj_to_spline.py, coefficients to ~1e-9 relative) in O(m + n) parallel work per pass.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "ba_math.h"

namespace mvus {

// Double-double arithmetic (~32 digits) for the passes whose normal equations are too ill conditioned for fp64: a knot set
// with almost as many knots as samples can push cond(A) to 1e10 (cond(A^T A) = 1e20); FITPACK's Givens rotations work on A
// itself.  Products of doubles are exact (FMA), sums carry their rounding error.
// The error-free transformations below are only error free when every operation rounds on its own: no fusing of a product
// with a neighbouring sum (hipcc contracts by default).
#if defined(__clang__)
#define MVUS_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define MVUS_NO_CONTRACT
#endif
struct dd {
  double hi, lo;
  dd() = default;
  MVUS_HD dd(double h) : hi(h), lo(0.0) {}
  MVUS_HD dd(double h, double l) : hi(h), lo(l) {}
};
MVUS_HD dd dd_quick(double a, double b) { MVUS_NO_CONTRACT const double s = a + b; return dd(s, b - (s - a)); }
MVUS_HD dd dd_two_sum(double a, double b) { MVUS_NO_CONTRACT const double s = a + b, bb = s - a; return dd(s, (a - (s - bb)) + (b - bb)); }
MVUS_HD dd dd_two_prod(double a, double b) { MVUS_NO_CONTRACT const double p = a * b; return dd(p, fma(a, b, -p)); }
MVUS_HD dd operator+(dd a, dd b) { MVUS_NO_CONTRACT dd s = dd_two_sum(a.hi, b.hi); s.lo += a.lo + b.lo; return dd_quick(s.hi, s.lo); }
MVUS_HD dd operator-(dd a) { return dd(-a.hi, -a.lo); }
MVUS_HD dd operator-(dd a, dd b) { return a + (-b); }
MVUS_HD dd operator*(dd a, dd b) { MVUS_NO_CONTRACT dd p = dd_two_prod(a.hi, b.hi); p.lo += a.hi * b.lo + a.lo * b.hi; return dd_quick(p.hi, p.lo); }
MVUS_HD dd operator/(dd a, dd b) {
  MVUS_NO_CONTRACT
  const double q1 = a.hi / b.hi;
  dd r = a - b * dd(q1);
  const double q2 = r.hi / b.hi;
  r = r - b * dd(q2);
  const double q3 = r.hi / b.hi;
  return dd_quick(q1, q2) + dd(q3);
}
MVUS_HD dd& operator+=(dd& a, dd b) { a = a + b; return a; }
MVUS_HD dd& operator-=(dd& a, dd b) { a = a - b; return a; }
MVUS_HD dd& operator/=(dd& a, dd b) { a = a / b; return a; }
MVUS_HD dd num_sqrt(dd a) {                        // Karp / Markstein: one correction of the fp64 square root
  MVUS_NO_CONTRACT
  const double x = 1.0 / sqrt(a.hi), ax = a.hi * x;
  const dd r = a - dd_two_prod(ax, ax);
  return dd_two_sum(ax, r.hi * (x * 0.5));
}
MVUS_HD double num_sqrt(double a) { return sqrt(a); }
MVUS_HD double to_double(double a) { return a; }
MVUS_HD double to_double(dd a) { return a.hi + a.lo; }
MVUS_HD bool is_positive(double a) { return a > 0.0; }
MVUS_HD bool is_positive(dd a) { return a.hi > 0.0; }
MVUS_HD double pivot_floor(double) { return 1e-12; }       // a Cholesky pivot below this fraction of its diagonal entry is lost to rounding
MVUS_HD double pivot_floor(dd) { return 1e-24; }
MVUS_HD double num_prod(double a, double b, double) { return a * b; }          // a * b in the precision of the third argument
MVUS_HD dd num_prod(double a, double b, dd) { return dd_two_prod(a, b); }

struct BandParts {
  int P = 1, len = 0, rem = 0, HB = 3;       // interior p: rows [first(p), first(p) + length(p)), then HB separator rows
  MVUS_HD int length(int p) const { return len + (p < rem ? 1 : 0); }
  MVUS_HD int first(int p) const { return p * (len + HB) + (p < rem ? p : rem); }
  // row j -> its interior p and the row i inside it (i >= length(p): a separator row)
  MVUS_HD void locate(int j, int& p, int& i) const {
    const int big = len + 1 + HB;
    if (j < rem * big) { p = j / big; i = j - p * big; }
    else { const int j2 = j - rem * big; p = rem + j2 / (len + HB); i = j2 - (p - rem) * (len + HB); }
  }
  // the interiors' rows are also kept TRANSPOSED for the solver: value k of K per row, row i of interior p at ((i K + k) P + p),
  // so that the P lanes of k_band_solve_parts (one interior each) read and write consecutive addresses
  MVUS_HD long long at(int p, int i, int k, int K) const { return ((long long)i * K + k) * P + p; }
};
inline BandParts band_parts(int n, int HB, int min_rows) {
  BandParts bp;
  bp.HB = HB;
  if (n < min_rows) return bp;
  static const double scale = [] { const char* e = std::getenv("MVUS_BAND_PARTS_SCALE"); return e ? std::atof(e) : 1.0; }();      // (experiments)
  int P = (int)(scale * std::sqrt((double)n / 2.5));
  P = std::max(2, std::min(256, P));
  while (P > 1 && (n - (P - 1) * HB) / P < 2 * HB) --P;
  if (P < 2) return bp;
  bp.P = P;
  bp.len = (n - (P - 1) * HB) / P;
  bp.rem = (n - (P - 1) * HB) - bp.len * P;
  return bp;
}
#if defined(__HIPCC__)
constexpr int kFitBlk = 22;      // per span: 10 products h_a h_b (a >= b) + 12 products h_a x_d


__global__ __launch_bounds__(256) void k_fit_basis(long long m, const double* __restrict__ u, const double* __restrict__ t, int ncoef,
                                                   int32_t* __restrict__ span, double* __restrict__ q) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= m) return;
  const double x = u[i];
  const int l = find_span(t, ncoef, x);
  double h[4], dh[4];
  bspline_basis<false>(t, l, x, h, dh);
  span[i] = l - 3;
#pragma unroll
  for (int a = 0; a < 4; ++a) q[4 * i + a] = h[a];
}

// first sample of every span (samples are sorted): first[sp] = lower_bound(u, t[3 + sp]), first[nspan] = m
__global__ __launch_bounds__(256) void k_fit_first(long long m, const double* __restrict__ u, const double* __restrict__ t, int nspan,
                                                   long long* __restrict__ first) {
  const int sp = blockIdx.x * 256 + threadIdx.x;
  if (sp > nspan) return;
  if (sp == nspan) { first[sp] = m; return; }
  if (sp == 0) { first[0] = 0; return; }
  const double knot = t[3 + sp];
  long long lo = 0, hi = m;                          // first index with u >= knot
  while (lo < hi) { const long long mid = (lo + hi) >> 1; if (u[mid] < knot) lo = mid + 1; else hi = mid; }
  first[sp] = lo;
}

// fixed-order sum of NV values per thread over one workgroup of NT -> thread 0 holds the result in v[]
template <int NV, int NT = 256, class T = double>
__device__ __forceinline__ void block_sum(T (&v)[NV], T* lds) {
  const int tid = threadIdx.x;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) lds[k * NT + tid] = v[k];
  __syncthreads();
  for (int off = NT / 2; off > 0; off >>= 1) {
    if (tid < off) {
#pragma unroll
      for (int k = 0; k < NV; ++k) lds[k * NT + tid] = lds[k * NT + tid] + lds[k * NT + tid + off];
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = lds[k * NT];
}

// A span with many samples (the first passes of a fit: ONE span holds all 3.3 M samples of a long trajectory) is cut into
// gridDim.y slices, one workgroup each; slice `sl` of span `sp` goes to block sl * nspan + sp of SB and k_fit_slice_sum adds the
// slices in their order.  fit_slices() picks the count from the number of spans alone, so the order of every sum is fixed.
inline int fit_slices(int nspan) {
  static const int cap = [] { const char* e = std::getenv("MVUS_FIT_SLICES_MAX"); return e ? std::max(1, std::min(256, std::atoi(e))) : 256; }();
  return nspan >= 512 ? 1 : std::min(cap, (1024 + nspan - 1) / nspan);
}
constexpr int kFitSliceBlocks = 1536;              // nspan * fit_slices(nspan) - nspan stays below this
template <class T, int NT>
__global__ __launch_bounds__(NT) void k_fit_blocks(long long m, const long long* __restrict__ first, const double* __restrict__ q,
                                                   const double* __restrict__ X, T* __restrict__ SB) {
  __shared__ T lds[kFitBlk * NT];
  const int sp = blockIdx.x;
  T v[kFitBlk];
#pragma unroll
  for (int k = 0; k < kFitBlk; ++k) v[k] = T(0.0);
  const long long span_lo = first[sp], span_hi = first[sp + 1];
  const long long per = (span_hi - span_lo + gridDim.y - 1) / gridDim.y;
  const long long lo = span_lo + per * blockIdx.y, hi = lo + per < span_hi ? lo + per : span_hi;
  for (long long i = lo + threadIdx.x; i < hi; i += NT) {
    double h[4], x[3];
#pragma unroll
    for (int a = 0; a < 4; ++a) h[a] = q[4 * i + a];
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = X[(long long)d * m + i];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int b = 0; b <= a; ++b) v[a * (a + 1) / 2 + b] += num_prod(h[a], h[b], T());
#pragma unroll
      for (int d = 0; d < 3; ++d) v[10 + 3 * a + d] += num_prod(h[a], x[d], T());
    }
  }
  block_sum<kFitBlk, NT, T>(v, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < kFitBlk; ++k) SB[((long long)blockIdx.y * gridDim.x + sp) * kFitBlk + k] = v[k];
  }
}

// the slices of k_fit_blocks added up in their order, into slice 0 (count = nspan * kFitBlk values per slice)
template <class T>
__global__ __launch_bounds__(256) void k_fit_slice_sum(long long count, int nslice, T* __restrict__ SB) {
  const long long e = blockIdx.x * 256ll + threadIdx.x;
  if (e >= count) return;
  T v = SB[e];
#pragma unroll 8
  for (int sl = 1; sl < nslice; ++sl) v += SB[(long long)sl * count + e];
  SB[e] = v;
}

// lower banded storage, width W = HB + 1: G[j * W + w] = M(j, j - w).  Here the normal equations in a width-5 array (w = 4
// zero) so that the penalty can be added in place, and rhs[d * ncoef + j].
template <class T>
__global__ __launch_bounds__(256) void k_fit_band(int ncoef, int nspan, const T* __restrict__ SB, T* __restrict__ G5, T* __restrict__ rhs, BandParts bp,
                                                  T* __restrict__ Mt, T* __restrict__ rt) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= ncoef) return;
  T g[5] = {T(0.0), T(0.0), T(0.0), T(0.0), T(0.0)}, r[3] = {T(0.0), T(0.0), T(0.0)};
  for (int sp = max(0, j - 3); sp <= min(nspan - 1, j); ++sp) {          // spans whose four coefficients include j
    const int a = j - sp;
    const T* blk = SB + (long long)sp * kFitBlk;
#pragma unroll
    for (int w = 0; w < 4; ++w) if (a - w >= 0) g[w] += blk[a * (a + 1) / 2 + (a - w)];
#pragma unroll
    for (int d = 0; d < 3; ++d) r[d] += blk[10 + 3 * a + d];
  }
#pragma unroll
  for (int w = 0; w < 5; ++w) G5[5 * (long long)j + w] = g[w];
#pragma unroll
  for (int d = 0; d < 3; ++d) rhs[(long long)d * ncoef + j] = r[d];
  if (bp.P > 1) {                                    // the copy k_band_solve_parts<3> reads (see BandParts::at)
    int p, i;
    bp.locate(j, p, i);
    if (i < bp.length(p)) {
#pragma unroll
      for (int w = 0; w < 4; ++w) Mt[bp.at(p, i, w, 4)] = g[w];
#pragma unroll
      for (int d = 0; d < 3; ++d) rt[bp.at(p, i, d, 3)] = r[d];
    }
  }
}

// BtB[j * 5 + w] = (B^T B)(j, j - w) for the n8 x ncoef jump matrix B whose row `it` holds b[it * 5 + c] in column it + c
template <class T>
__global__ __launch_bounds__(256) void k_fit_penalty(int ncoef, int n8, const double* __restrict__ b, T* __restrict__ BtB) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= ncoef) return;
  T g[5] = {T(0.0), T(0.0), T(0.0), T(0.0), T(0.0)};
  for (int it = max(0, j - 4); it <= min(n8 - 1, j); ++it) {
    const int cj = j - it;
#pragma unroll
    for (int w = 0; w < 5; ++w) if (cj - w >= 0) g[w] += num_prod(b[5 * (long long)it + cj], b[5 * (long long)it + cj - w], T());
  }
#pragma unroll
  for (int w = 0; w < 5; ++w) BtB[5 * (long long)j + w] = g[w];
}

template <class T>
__global__ __launch_bounds__(256) void k_fit_combine(long long count, const T* __restrict__ G5, const T* __restrict__ BtB, double pinv, T* __restrict__ M,
                                                     int ncoef, const T* __restrict__ rhs, BandParts bp, T* __restrict__ Mt, T* __restrict__ rt) {
  const long long e = blockIdx.x * 256ll + threadIdx.x;
  if (e >= count) return;
  const T v = G5[e] + (T(pinv) * T(pinv)) * BtB[e];                  // fppara rotates the rows of B in with weight 1/p
  M[e] = v;
  if (bp.P > 1) {                                    // the copy k_band_solve_parts<4> reads, and the right-hand sides in ITS partition
    const int j = (int)(e / 5), w = (int)(e - 5ll * j);
    int p, i;
    bp.locate(j, p, i);
    if (i < bp.length(p)) {
      Mt[bp.at(p, i, w, 5)] = v;
      if (w < 3) rt[bp.at(p, i, w, 3)] = rhs[(long long)w * ncoef + j];
    }
  }
}

// Banded Cholesky M = L L^T (lower band, row stride 5, HB = 3 or 4 sub-diagonals used) and the solves for the three
// right-hand sides, ONE wavefront: every lane runs the factor recurrence (no exchange between lanes), the previous HB rows
// of L stay in registers, the next row of M is fetched while the current one is processed; lanes 0..2 carry one right-hand
// side each.  L (stride 5) and the solution go to Lout / c; out[0] = sum of the diagonal of L (FITPACK's sum of a(i,1)).
template <int HB, class T>
__global__ __launch_bounds__(64) void k_band_solve(int n, const T* __restrict__ M, const T* __restrict__ rhs, T* __restrict__ Lout, T* __restrict__ ywork,
                                                   double* __restrict__ c, double* __restrict__ out, int* __restrict__ fail) {
  const int lane = threadIdx.x;
  const int d = lane < 3 ? lane : 0;
  const T* bvec = rhs + (long long)d * n;
  T* yv = ywork + (long long)d * n;
  double* cv = c + (long long)d * n;
  T Lp[HB + 1][HB + 1];               // Lp[u][w] = L(j-u, j-u-w), u = 1..HB
  T yp[HB + 1];                       // yp[u] = y(j-u)
#pragma unroll
  for (int u2 = 0; u2 <= HB; ++u2) {
    yp[u2] = T(0.0);
#pragma unroll
    for (int w = 0; w <= HB; ++w) Lp[u2][w] = (w == 0) ? T(1.0) : T(0.0);
  }
  T nxt[HB + 1], nb = T(0.0);
#pragma unroll
  for (int w = 0; w <= HB; ++w) nxt[w] = n > 0 ? M[w] : T(0.0);
  nb = n > 0 ? bvec[0] : T(0.0);
  double dsum = 0.0, dmin = 1e300, dmax = 0.0;
  bool bad = false;
  for (int j = 0; j < n; ++j) {
    T row[HB + 1];
#pragma unroll
    for (int w = 0; w <= HB; ++w) row[w] = nxt[w];
    const T bj = nb;
    if (j + 1 < n) {
#pragma unroll
      for (int w = 0; w <= HB; ++w) nxt[w] = M[5 * (long long)(j + 1) + w];
      nb = bvec[j + 1];
    }
#pragma unroll
    for (int w = HB; w >= 1; --w) {                 // L(j, j-w)
      T v = row[w];
#pragma unroll
      for (int u2 = w + 1; u2 <= HB; ++u2) v -= row[u2] * Lp[w][u2 - w];
      row[w] = (j - w >= 0) ? v / Lp[w][0] : T(0.0);
    }
    // A pivot that cancels to (almost) nothing means a direction the samples do not determine: FITPACK's knot search does
    // produce such knot sets close to interpolation, its Givens triangle then holds a rounding-noise diagonal entry.  The
    // residual does not depend on how that direction is resolved (the normal equations are consistent), so the pivot is
    // floored instead of failing; the caller repeats an fp64 pass with floored pivots in double-double.
    const T floor_ = T(pivot_floor(T()) * to_double(row[0]));
    T dg = row[0];
#pragma unroll
    for (int u2 = 1; u2 <= HB; ++u2) dg -= row[u2] * row[u2];
    const bool lost = !(to_double(dg) > to_double(floor_));
    bad |= lost;
    dg = lost ? (is_positive(floor_) ? floor_ : T(1.0)) : dg;
    row[0] = num_sqrt(dg);
    const double dl = to_double(row[0]);
    dsum += dl; dmin = fmin(dmin, dl); dmax = fmax(dmax, dl);
    T y = bj;
#pragma unroll
    for (int u2 = 1; u2 <= HB; ++u2) y -= row[u2] * yp[u2];
    y /= row[0];
    if (lane == 0) {
#pragma unroll
      for (int w = 0; w <= HB; ++w) Lout[5 * (long long)j + w] = row[w];
    }
    if (lane < 3) yv[j] = y;
#pragma unroll
    for (int u2 = HB; u2 >= 2; --u2) {
      yp[u2] = yp[u2 - 1];
#pragma unroll
      for (int w = 0; w <= HB; ++w) Lp[u2][w] = Lp[u2 - 1][w];
    }
    yp[1] = y;
#pragma unroll
    for (int w = 0; w <= HB; ++w) Lp[1][w] = row[w];
  }
  if (lane == 0) { out[0] = dsum; out[2] = dmin; out[3] = dmax; if (bad) fail[0] = 1; }
  __syncthreads();                                   // Lout / y written by this wavefront are read back below
  __threadfence_block();
  // L^T c = y: c(j) = (y(j) - sum_u L(j+u, j) c(j+u)) / L(j,j)
  T Ln[HB + 1][HB + 1];               // Ln[u][w] = L(j+u, j+u-w)
  T cn[HB + 1];
#pragma unroll
  for (int u2 = 0; u2 <= HB; ++u2) {
    cn[u2] = T(0.0);
#pragma unroll
    for (int w = 0; w <= HB; ++w) Ln[u2][w] = T(0.0);
  }
  for (int j = n - 1; j >= 0; --j) {
    T row[HB + 1];
#pragma unroll
    for (int w = 0; w <= HB; ++w) row[w] = Lout[5 * (long long)j + w];
    T v = lane < 3 ? yv[j] : T(0.0);
#pragma unroll
    for (int u2 = 1; u2 <= HB; ++u2) v -= Ln[u2][u2] * cn[u2];
    v /= row[0];
    if (lane < 3) cv[j] = to_double(v);
#pragma unroll
    for (int u2 = HB; u2 >= 2; --u2) {
      cn[u2] = cn[u2 - 1];
#pragma unroll
      for (int w = 0; w <= HB; ++w) Ln[u2][w] = Ln[u2 - 1][w];
    }
    cn[1] = v;
#pragma unroll
    for (int w = 0; w <= HB; ++w) Ln[1][w] = row[w];
  }
}

#endif
// ---- the same systems, partitioned --------------------------------------------------------------------------------------
// k_band_solve is one dependent chain of n rows (0.4 us a row in fp64, 4.8 us in double-double: 31 ms for the 6 500 coefficients
// of a 3.3 M-sample trajectory, and ~50 such solves per fit).  k_band_solve_parts cuts the rows into P interiors separated by
// P - 1 separators of HB rows (interiors do not couple: half-bandwidth <= HB), one LANE per interior:
//   phase 1  every lane factors its interior A_p = L L^T and forward-solves the coupling columns to the separator on its left
//            (Y_L, dense), to the one on its right (Y_R, last HB rows only) and the three right-hand sides (y); what the
//            interior contributes to the separator system (Y^T Y, Y^T y) goes to a record per interior
//   phase 2  one lane: block-tridiagonal Cholesky of the separator system (HB x HB blocks), both substitutions
//   phase 3  every lane: L^T x = y - Y_L x_left - Y_R x_right
// One workgroup (P <= 256 lanes), __syncthreads between the phases.  The elimination ORDER differs from k_band_solve's, so
// the factor's diagonal is not FITPACK's a(i,1): out[0] is not written; the caller runs k_band_solve once per fit where
// fppara needs that sum (the initial p).  out[2], out[3] = smallest / largest pivot of all the factors (same use as before).
MVUS_HD double num_recip(double a) { return 1.0 / a; }
MVUS_HD dd num_recip(dd a) {                        // one Newton step on the fp64 reciprocal: r0 + r0 (1 - a r0)
  MVUS_NO_CONTRACT
  const double r0 = 1.0 / a.hi;
  dd p = dd_two_prod(a.hi, r0);
  p.lo += a.lo * r0;
  const dd e = dd(1.0) - dd_quick(p.hi, p.lo);
  return dd_two_sum(r0, r0 * (e.hi + e.lo));
}

#if defined(__HIPCC__)
template <int HB> struct BandRec {                  // per interior (T units)
  static constexpr int GLL = 0, GRR = HB * HB, C = 2 * HB * HB, YR = 3 * HB * HB, GL = 4 * HB * HB, GR = 4 * HB * HB + 3 * HB, SIZE = 4 * HB * HB + 6 * HB;
};
template <int HB> struct BandSep {                  // per separator (T units): its Cholesky block, the coupling W to the next one, z, x
  static constexpr int LT = 0, W = HB * HB, Z = 2 * HB * HB, X = 2 * HB * HB + 3 * HB, SIZE = 2 * HB * HB + 6 * HB;
};
constexpr int kBandPartsMax = 256;
constexpr int kBandPartsWork = kBandPartsMax * (BandRec<4>::SIZE + BandSep<4>::SIZE);        // T units of `work`

template <class T>
struct PivotStats {
  double dsum = 0.0, dmin = 1e300, dmax = 0.0;
  bool bad = false;
  // dg = what is left of the diagonal entry `full` after the eliminations: the pivot (floored when lost to rounding, see k_band_solve)
  __device__ __forceinline__ T pivot(T dg, T full) {
    const T floor_ = T(pivot_floor(T()) * to_double(full));
    const bool lost = !(to_double(dg) > to_double(floor_));
    bad |= lost;
    dg = lost ? (is_positive(floor_) ? floor_ : T(1.0)) : dg;
    const T l = num_sqrt(dg);
    const double dl = to_double(l);
    dsum += dl; dmin = fmin(dmin, dl); dmax = fmax(dmax, dl);
    return l;
  }
};

// sum of diag(L) of the banded Cholesky in the NATURAL order (FITPACK's sum of a(i,1), the initial p of the smoothing
// iteration) -- the factor recurrence of k_band_solve alone, with reciprocal pivots; every lane fetches one row of a 64-row
// chunk (coalesced, the next chunk while this one is processed) and the rows are broadcast from the lanes' registers.
__device__ __forceinline__ double lane_value(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ dd lane_value(dd v, int lane) { return dd(lane_value(v.hi, lane), lane_value(v.lo, lane)); }
template <int HB, class T>
__global__ __launch_bounds__(64) void k_band_diag_sum(int n, const T* __restrict__ M, double* __restrict__ out) {
  const int lane = threadIdx.x;
  T Lp[HB + 1][HB + 1], ip[HB + 1];
#pragma unroll
  for (int u = 0; u <= HB; ++u) {
    ip[u] = T(1.0);
#pragma unroll
    for (int w = 0; w <= HB; ++w) Lp[u][w] = (w == 0) ? T(1.0) : T(0.0);
  }
  PivotStats<T> ps;
  T cur[HB + 1], nxt[HB + 1];
#pragma unroll
  for (int w = 0; w <= HB; ++w) { cur[w] = lane < n ? M[5 * (long long)lane + w] : T(0.0); nxt[w] = T(0.0); }
  for (int base = 0; base < n; base += 64) {
    if (base + 64 + lane < n) {
#pragma unroll
      for (int w = 0; w <= HB; ++w) nxt[w] = M[5 * (long long)(base + 64 + lane) + w];
    }
    const int cnt = n - base < 64 ? n - base : 64;
    for (int k = 0; k < cnt; ++k) {
      T row[HB + 1];
#pragma unroll
      for (int w = 0; w <= HB; ++w) row[w] = lane_value(cur[w], k);
#pragma unroll
      for (int w = HB; w >= 1; --w) {
        T v = row[w];
#pragma unroll
        for (int u2 = w + 1; u2 <= HB; ++u2) v -= row[u2] * Lp[w][u2 - w];
        row[w] = (base + k - w >= 0) ? v * ip[w] : T(0.0);
      }
      const T full = row[0];
      T dg = row[0];
#pragma unroll
      for (int u2 = 1; u2 <= HB; ++u2) dg -= row[u2] * row[u2];
      row[0] = ps.pivot(dg, full);
#pragma unroll
      for (int u2 = HB; u2 >= 2; --u2) {
        ip[u2] = ip[u2 - 1];
#pragma unroll
        for (int w = 0; w <= HB; ++w) Lp[u2][w] = Lp[u2 - 1][w];
      }
      ip[1] = num_recip(row[0]);
#pragma unroll
      for (int w = 0; w <= HB; ++w) Lp[1][w] = row[w];
    }
#pragma unroll
    for (int w = 0; w <= HB; ++w) cur[w] = nxt[w];
  }
  if (lane == 0) out[0] = ps.dsum;
}

template <int HB, class T>
__global__ __launch_bounds__(kBandPartsMax) void k_band_solve_parts(int n, BandParts bp, const T* __restrict__ M, const T* __restrict__ rhs, T* __restrict__ Mt,
                                                                    T* __restrict__ rt, T* __restrict__ YL, T* __restrict__ work, double* __restrict__ c,
                                                                    double* __restrict__ out, int* __restrict__ fail) {
  // M, rhs: the system as the other kernels hold it (separator rows are read from there); Mt, rt: the interiors' rows transposed
  // (written by k_fit_band / k_fit_combine) -- overwritten in place by L and y, then by nothing: x goes straight to c
  constexpr int NW = HB + 1;
  using R = BandRec<HB>;
  using S = BandSep<HB>;
  constexpr int NC = HB + 3;                          // forward columns of an interior: HB couplings to the left, 3 right-hand sides
  const int p = threadIdx.x, P = bp.P;
  T* rec = work;
  T* sep = work + (long long)kBandPartsMax * R::SIZE;
  __shared__ double s_min[kBandPartsMax], s_max[kBandPartsMax];
  __shared__ int s_bad[kBandPartsMax];
  PivotStats<T> ps;
  const int r0 = p < P ? bp.first(p) : 0, lp = p < P ? bp.length(p) : 0;
  // ---- phase 1 ----
  if (p < P) {
    T Lp[HB + 1][HB + 1];                             // Lp[u][w] = L(i-u, i-u-w)
    T ip[HB + 1];                                     // ip[u] = 1 / L(i-u, i-u)
    T Yp[HB + 1][NC];                                 // Yp[u][col] = Y(i-u, col)
    T gll[HB][HB], gl[HB][3];
#pragma unroll
    for (int u = 0; u <= HB; ++u) {
      ip[u] = T(1.0);
#pragma unroll
      for (int w = 0; w <= HB; ++w) Lp[u][w] = (w == 0) ? T(1.0) : T(0.0);
#pragma unroll
      for (int col = 0; col < NC; ++col) Yp[u][col] = T(0.0);
    }
#pragma unroll
    for (int a = 0; a < HB; ++a) {
#pragma unroll
      for (int b = 0; b < HB; ++b) gll[a][b] = T(0.0);
#pragma unroll
      for (int d = 0; d < 3; ++d) gl[a][d] = T(0.0);
    }
    // rows are fetched RB at a time (all loads of a batch in flight together): with one row ahead every row waited a full memory
    // round trip (~0.7 us against ~0.15 us of fp64 arithmetic)
    constexpr int RB = sizeof(T) == sizeof(double) ? 8 : 2;
    for (int i0 = 0; i0 < lp; i0 += RB) {
     T cm[RB][HB + 1], cb[RB][3];
#pragma unroll
     for (int r = 0; r < RB; ++r) {
       const int ii = i0 + r < lp ? i0 + r : lp - 1;
#pragma unroll
       for (int w = 0; w <= HB; ++w) cm[r][w] = Mt[bp.at(p, ii, w, NW)];
#pragma unroll
       for (int d = 0; d < 3; ++d) cb[r][d] = rt[bp.at(p, ii, d, 3)];
     }
#pragma unroll
     for (int r = 0; r < RB; ++r) {
      const int i = i0 + r;
      if (i >= lp) break;
      T row[HB + 1], b[NC];
#pragma unroll
      for (int w = 0; w <= HB; ++w) row[w] = cm[r][w];
#pragma unroll
      for (int col = 0; col < HB; ++col) b[col] = T(0.0);
#pragma unroll
      for (int d = 0; d < 3; ++d) b[HB + d] = cb[r][d];
      if (i < HB) {                                   // M(j, j-w) with i - w < 0 couples to the separator on the left: column i + HB - w of it
#pragma unroll
        for (int w = 1; w <= HB; ++w) {
          if (w > i) {
#pragma unroll
            for (int col = 0; col < HB; ++col) if (col == i + HB - w) b[col] = row[w];
            row[w] = T(0.0);
          }
        }
      }
#pragma unroll
      for (int w = HB; w >= 1; --w) {                 // L(i, i-w)
        T v = row[w];
#pragma unroll
        for (int u2 = w + 1; u2 <= HB; ++u2) v -= row[u2] * Lp[w][u2 - w];
        row[w] = v * ip[w];
      }
      const T full = row[0];
      T dg = row[0];
#pragma unroll
      for (int u2 = 1; u2 <= HB; ++u2) dg -= row[u2] * row[u2];
      row[0] = ps.pivot(dg, full);
      const T inv = num_recip(row[0]);
      T y[NC];
#pragma unroll
      for (int col = 0; col < NC; ++col) {
        T v = b[col];
#pragma unroll
        for (int u2 = 1; u2 <= HB; ++u2) v -= row[u2] * Yp[u2][col];
        y[col] = v * inv;
      }
#pragma unroll
      for (int a = 0; a < HB; ++a) {
#pragma unroll
        for (int b2 = 0; b2 <= a; ++b2) gll[a][b2] += y[a] * y[b2];
#pragma unroll
        for (int d = 0; d < 3; ++d) gl[a][d] += y[a] * y[HB + d];
      }
#pragma unroll
      for (int w = 0; w <= HB; ++w) Mt[bp.at(p, i, w, NW)] = row[w];
#pragma unroll
      for (int col = 0; col < HB; ++col) YL[bp.at(p, i, col, HB)] = y[col];
#pragma unroll
      for (int d = 0; d < 3; ++d) rt[bp.at(p, i, d, 3)] = y[HB + d];
#pragma unroll
      for (int u2 = HB; u2 >= 2; --u2) {
        ip[u2] = ip[u2 - 1];
#pragma unroll
        for (int w = 0; w <= HB; ++w) Lp[u2][w] = Lp[u2 - 1][w];
#pragma unroll
        for (int col = 0; col < NC; ++col) Yp[u2][col] = Yp[u2 - 1][col];
      }
      ip[1] = inv;
#pragma unroll
      for (int w = 0; w <= HB; ++w) Lp[1][w] = row[w];
#pragma unroll
      for (int col = 0; col < NC; ++col) Yp[1][col] = y[col];
     }
    }
    T* rc = rec + (long long)p * R::SIZE;
#pragma unroll
    for (int a = 0; a < HB; ++a) {
#pragma unroll
      for (int b2 = 0; b2 < HB; ++b2) rc[R::GLL + a * HB + b2] = b2 <= a ? gll[a][b2] : gll[b2][a];
#pragma unroll
      for (int d = 0; d < 3; ++d) rc[R::GL + a * 3 + d] = gl[a][d];
    }
    if (p + 1 < P) {
      // separator row r0 + lp + cq couples to the interior row lp - HB + r (window index u = HB - r) through M(., w = HB - r + cq):
      // Y_R(r, cq), lower triangular in (r, cq)
      const long long s0 = (long long)r0 + lp;
      T yr[HB][HB];
#pragma unroll
      for (int r = 0; r < HB; ++r) {
#pragma unroll
        for (int cq = 0; cq < HB; ++cq) {
          if (cq <= r) {
            T v = M[5 * (s0 + cq) + (HB - r + cq)];
#pragma unroll
            for (int r2 = 0; r2 < HB; ++r2) if (r2 >= cq && r2 < r) v -= Lp[HB - r][r - r2] * yr[r2][cq];
            yr[r][cq] = v * ip[HB - r];
          } else {
            yr[r][cq] = T(0.0);
          }
        }
      }
#pragma unroll
      for (int a = 0; a < HB; ++a) {
#pragma unroll
        for (int b2 = 0; b2 < HB; ++b2) {
          T grr = T(0.0), cc = T(0.0);
#pragma unroll
          for (int r = 0; r < HB; ++r) { grr += yr[r][a] * yr[r][b2]; cc += Yp[HB - r][a] * yr[r][b2]; }
          rc[R::GRR + a * HB + b2] = grr;
          rc[R::C + a * HB + b2] = cc;                // C(a, b2) = sum_i Y_L(i, a) Y_R(i, b2)
          rc[R::YR + a * HB + b2] = yr[a][b2];
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          T g = T(0.0);
#pragma unroll
          for (int r = 0; r < HB; ++r) g += yr[r][a] * Yp[HB - r][HB + d];
          rc[R::GR + a * 3 + d] = g;
        }
      }
    }
  }
  __syncthreads();
  __threadfence_block();
  // ---- phase 2: the separator system: lanes 0..2 of the first wavefront, one right-hand side each (the matrix part is the same
  // arithmetic in all three; lane 0 stores it).  The inputs of separator q + 1 are fetched while separator q is processed: a step
  // was 2.6 us (fp64), one memory round trip of it waiting for ~60 loads.
  if (p < 3) {
    const int dcol = p;
    struct SepIn { T grr[HB][HB], gll[HB][HB], cm[HB][HB], m[HB][HB], gr[HB], gl[HB], rh[HB]; };
    auto load_sep = [&](int q, SepIn& in) {
      const long long s0 = (long long)bp.first(q) + bp.length(q);
      const T* ra = rec + (long long)q * R::SIZE;
      const T* rb = rec + (long long)(q + 1) * R::SIZE;
#pragma unroll
      for (int a = 0; a < HB; ++a) {
#pragma unroll
        for (int b2 = 0; b2 < HB; ++b2) {
          in.grr[a][b2] = b2 <= a ? ra[R::GRR + a * HB + b2] : T(0.0);
          in.gll[a][b2] = b2 <= a ? rb[R::GLL + a * HB + b2] : T(0.0);
          in.cm[a][b2] = rb[R::C + a * HB + b2];
          in.m[a][b2] = b2 <= a ? M[5 * (s0 + a) + (a - b2)] : T(0.0);
        }
        in.gr[a] = ra[R::GR + a * 3 + dcol];
        in.gl[a] = rb[R::GL + a * 3 + dcol];
        in.rh[a] = rhs[(long long)dcol * n + s0 + a];
      }
    };
    T W[HB][HB], z[HB];
    PivotStats<T> ps2;                                // the separator pivots (lane 0 adds them to its statistics)
    // (fp64 only: in double-double the second set of blocks does not fit the register file -- 312 B of scratch per lane -- and the
    // arithmetic of a step, ~12 us, hides nothing worth hiding)
    constexpr bool kAhead = sizeof(T) == sizeof(double);
    SepIn nxt_in;
    if (kAhead && P > 1) load_sep(0, nxt_in);
    for (int q = 0; q + 1 < P; ++q) {
      SepIn in;
      if (kAhead) { in = nxt_in; if (q + 2 < P) load_sep(q + 1, nxt_in); }
      else load_sep(q, in);
      T D[HB][HB], full[HB], r[HB];
#pragma unroll
      for (int a = 0; a < HB; ++a) {
#pragma unroll
        for (int b2 = 0; b2 <= a; ++b2) {
          T v = in.m[a][b2] - in.grr[a][b2] - in.gll[a][b2];
          if (b2 == a) full[a] = in.m[a][a];
          if (q > 0) {
#pragma unroll
            for (int k = 0; k < HB; ++k) v -= W[a][k] * W[b2][k];
          }
          D[a][b2] = v;
        }
        T v = in.rh[a] - in.gr[a] - in.gl[a];
        if (q > 0) {
#pragma unroll
          for (int k = 0; k < HB; ++k) v -= W[a][k] * z[k];
        }
        r[a] = v;
      }
      T Lt[HB][HB], il[HB];
#pragma unroll
      for (int a = 0; a < HB; ++a) {
#pragma unroll
        for (int b2 = 0; b2 < a; ++b2) {
          T v = D[a][b2];
#pragma unroll
          for (int k = 0; k < HB; ++k) if (k < b2) v -= Lt[a][k] * Lt[b2][k];
          Lt[a][b2] = v * il[b2];
        }
        T dg = D[a][a];
#pragma unroll
        for (int k = 0; k < HB; ++k) if (k < a) dg -= Lt[a][k] * Lt[a][k];
        Lt[a][a] = ps2.pivot(dg, full[a]);
        il[a] = num_recip(Lt[a][a]);
        T v = r[a];
#pragma unroll
        for (int k = 0; k < HB; ++k) if (k < a) v -= Lt[a][k] * z[k];
        z[a] = v * il[a];
      }
      T* sq = sep + (long long)q * S::SIZE;
#pragma unroll
      for (int a = 0; a < HB; ++a) {
        if (dcol == 0) {
#pragma unroll
          for (int b2 = 0; b2 < HB; ++b2) sq[S::LT + a * HB + b2] = b2 < a ? Lt[a][b2] : (b2 == a ? il[a] : T(0.0));    // the diagonal holds 1 / L
        }
        sq[S::Z + a * 3 + dcol] = z[a];
      }
      if (q + 2 < P) {                                // W = O Lt^-T, O(cq, cc) = -C_{q+1}(cc, cq): rows = the next separator
#pragma unroll
        for (int a = 0; a < HB; ++a) {
#pragma unroll
          for (int k = 0; k < HB; ++k) {
            T v = -in.cm[k][a];
#pragma unroll
            for (int m2 = 0; m2 < HB; ++m2) if (m2 < k) v -= W[a][m2] * Lt[k][m2];
            W[a][k] = v * il[k];
          }
        }
        if (dcol == 0) {
#pragma unroll
          for (int a = 0; a < HB; ++a) {
#pragma unroll
            for (int k = 0; k < HB; ++k) sq[S::W + a * HB + k] = W[a][k];
          }
        }
      }
    }
    if (dcol == 0) { ps.dsum += ps2.dsum; ps.dmin = fmin(ps.dmin, ps2.dmin); ps.dmax = fmax(ps.dmax, ps2.dmax); ps.bad |= ps2.bad; }
    // lanes 1, 2 read what lane 0 stored (Lt, W): same wavefront, program order + the memory fence below
    __threadfence_block();
    T x[HB];
    struct SepBack { T lt[HB][HB], w[HB][HB], z[HB]; };
    auto load_back = [&](int q, SepBack& in) {
      const T* sq = sep + (long long)q * S::SIZE;
#pragma unroll
      for (int a = 0; a < HB; ++a) {
#pragma unroll
        for (int k = 0; k < HB; ++k) { in.lt[a][k] = sq[S::LT + a * HB + k]; in.w[a][k] = q + 2 < P ? sq[S::W + a * HB + k] : T(0.0); }
        in.z[a] = sq[S::Z + a * 3 + dcol];
      }
    };
    SepBack nxt_b;
    if (kAhead && P > 1) load_back(P - 2, nxt_b);
    for (int q = P - 2; q >= 0; --q) {                // x_q = Lt^-T (z_q - W_q^T x_{q+1}); the blocks of q - 1 are fetched meanwhile
      T* sq = sep + (long long)q * S::SIZE;
      SepBack in;
      if (kAhead) { in = nxt_b; if (q > 0) load_back(q - 1, nxt_b); }
      else load_back(q, in);
      T v[HB];
#pragma unroll
      for (int a = 0; a < HB; ++a) {
        T t = in.z[a];
        if (q + 2 < P) {
#pragma unroll
          for (int k = 0; k < HB; ++k) t -= in.w[k][a] * x[k];
        }
        v[a] = t;
      }
#pragma unroll
      for (int a = HB - 1; a >= 0; --a) {
        T t = v[a];
#pragma unroll
        for (int k = 0; k < HB; ++k) if (k > a) t -= in.lt[k][a] * x[k];
        x[a] = t * in.lt[a][a];
      }
      const long long s0 = (long long)bp.first(q) + bp.length(q);
#pragma unroll
      for (int a = 0; a < HB; ++a) { sq[S::X + a * 3 + dcol] = x[a]; c[(long long)dcol * n + s0 + a] = to_double(x[a]); }
    }
  }
  __syncthreads();
  __threadfence_block();
  // ---- phase 3 ----
  if (p < P) {
    T xl[HB][3], xr[HB][3], yr[HB][HB];
#pragma unroll
    for (int a = 0; a < HB; ++a) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        xl[a][d] = p > 0 ? sep[(long long)(p - 1) * S::SIZE + S::X + a * 3 + d] : T(0.0);
        xr[a][d] = p + 1 < P ? sep[(long long)p * S::SIZE + S::X + a * 3 + d] : T(0.0);
      }
#pragma unroll
      for (int b2 = 0; b2 < HB; ++b2) yr[a][b2] = p + 1 < P ? rec[(long long)p * R::SIZE + R::YR + a * HB + b2] : T(0.0);
    }
    T Ln[HB + 1][HB + 1], cn[HB + 1][3];              // Ln[u][w] = L(i+u, i+u-w), cn[u] = x(i+u)
#pragma unroll
    for (int u = 0; u <= HB; ++u) {
#pragma unroll
      for (int w = 0; w <= HB; ++w) Ln[u][w] = T(0.0);
#pragma unroll
      for (int d = 0; d < 3; ++d) cn[u][d] = T(0.0);
    }
    constexpr int RB = sizeof(T) == sizeof(double) ? 8 : 2;            // rows per batch of loads, as in phase 1
    for (int i1 = lp - 1; i1 >= 0; i1 -= RB) {
     T cm[RB][HB + 1], cyl[RB][HB], cy[RB][3];
#pragma unroll
     for (int r = 0; r < RB; ++r) {
       const int ii = i1 - r >= 0 ? i1 - r : 0;
#pragma unroll
       for (int w = 0; w <= HB; ++w) cm[r][w] = Mt[bp.at(p, ii, w, NW)];
#pragma unroll
       for (int col = 0; col < HB; ++col) cyl[r][col] = YL[bp.at(p, ii, col, HB)];
#pragma unroll
       for (int d = 0; d < 3; ++d) cy[r][d] = rt[bp.at(p, ii, d, 3)];
     }
#pragma unroll
     for (int r = 0; r < RB; ++r) {
      const int i = i1 - r;
      if (i < 0) break;
      const long long j = r0 + i;
      T row[HB + 1], yl[HB], v[3];
#pragma unroll
      for (int w = 0; w <= HB; ++w) row[w] = cm[r][w];
#pragma unroll
      for (int col = 0; col < HB; ++col) yl[col] = cyl[r][col];
#pragma unroll
      for (int d = 0; d < 3; ++d) v[d] = cy[r][d];
      const int rr = i - (lp - HB);                   // row of Y_R (>= 0 in the last HB rows of the interior)
      const T inv = num_recip(row[0]);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        T t = v[d];
#pragma unroll
        for (int col = 0; col < HB; ++col) t -= yl[col] * xl[col][d];
        if (rr >= 0) {
#pragma unroll
          for (int r = 0; r < HB; ++r) {
            if (r == rr) {
#pragma unroll
              for (int cq = 0; cq < HB; ++cq) t -= yr[r][cq] * xr[cq][d];
            }
          }
        }
#pragma unroll
        for (int u2 = 1; u2 <= HB; ++u2) t -= Ln[u2][u2] * cn[u2][d];
        t = t * inv;
        v[d] = t;
        c[(long long)d * n + j] = to_double(t);
      }
#pragma unroll
      for (int u2 = HB; u2 >= 2; --u2) {
#pragma unroll
        for (int w = 0; w <= HB; ++w) Ln[u2][w] = Ln[u2 - 1][w];
#pragma unroll
        for (int d = 0; d < 3; ++d) cn[u2][d] = cn[u2 - 1][d];
      }
#pragma unroll
      for (int w = 0; w <= HB; ++w) Ln[1][w] = row[w];
#pragma unroll
      for (int d = 0; d < 3; ++d) cn[1][d] = v[d];
     }
    }
  }
  s_min[p] = ps.dmin; s_max[p] = ps.dmax; s_bad[p] = ps.bad ? 1 : 0;
  __syncthreads();
  if (p == 0) {
    double mn = 1e300, mx = 0.0;
    int bad = 0;
    for (int k = 0; k < (int)blockDim.x; ++k) { mn = fmin(mn, s_min[k]); mx = fmax(mx, s_max[k]); bad |= s_bad[k]; }
    out[2] = mn; out[3] = mx;
    if (bad) fail[0] = 1;
  }
}
#endif

#if defined(__HIPCC__)
// squared residual of every sample, in fppara's order of operations (fac = sum_j c(j) q(it, j); term += (fac - x)^2)
__global__ __launch_bounds__(256) void k_fit_residual(long long m, int ncoef, const int32_t* __restrict__ span, const double* __restrict__ q,
                                                      const double* __restrict__ X, const double* __restrict__ c, double* __restrict__ term) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= m) return;
  const int sp = span[i];
  double tsum = 0.0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    double fac = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) fac = fac + c[(long long)d * ncoef + sp + a] * q[4 * i + a];
    const double r = fac - X[(long long)d * m + i];
    tsum = tsum + r * r;
  }
  term[i] = tsum;
}

// per-span residual: the first sample of a span (it is the first with u >= the knot that opens the span) is shared half / half
// with the span before, like fppara's `new` / `store` bookkeeping
// (gridDim.y slices per span as in k_fit_blocks: with more than one, the slice sums go to part[sl * nspan + sp] and
// k_fit_fpint_final adds them in order)
__global__ __launch_bounds__(256) void k_fit_fpint(int nspan, const long long* __restrict__ first, const double* __restrict__ term, double* __restrict__ fpint,
                                                   double* __restrict__ part) {
  __shared__ double lds[256];
  const int sp = blockIdx.x;
  const long long a = first[sp], b = first[sp + 1];
  const long long span_lo = a + (sp > 0 ? 1 : 0);
  const long long per = (b - span_lo + gridDim.y - 1) / gridDim.y;
  const long long lo = span_lo + per * blockIdx.y, hi = lo + per < b ? lo + per : b;
  double v[1] = {0.0};
  for (long long i = lo + threadIdx.x; i < hi; i += 256) v[0] += term[i];
  block_sum<1>(v, lds);
  if (threadIdx.x == 0) {
    if (gridDim.y > 1) { part[(long long)blockIdx.y * nspan + sp] = v[0]; return; }
    double s = v[0];
    if (sp > 0 && a < b) s += 0.5 * term[a];
    if (sp + 1 < nspan && first[sp + 1] < first[sp + 2]) s += 0.5 * term[b];
    fpint[sp] = s;
  }
}
__global__ __launch_bounds__(256) void k_fit_fpint_final(int nspan, int nslice, const long long* __restrict__ first, const double* __restrict__ term,
                                                         const double* __restrict__ part, double* __restrict__ fpint) {
  const int sp = blockIdx.x * 256 + threadIdx.x;
  if (sp >= nspan) return;
  const long long a = first[sp], b = first[sp + 1];
  double s = 0.0;
  for (int sl = 0; sl < nslice; ++sl) s += part[(long long)sl * nspan + sp];
  if (sp > 0 && a < b) s += 0.5 * term[a];
  if (sp + 1 < nspan && first[sp + 1] < first[sp + 2]) s += 0.5 * term[b];
  fpint[sp] = s;
}

// total of `count` values in a fixed order: per-workgroup partial sums (grid-stride), then one workgroup over the partials
// (k_fit_total with `count` = the number of partials).  One workgroup over 550k samples took 517 us a pass -- 65 % of a fit.
__global__ __launch_bounds__(256) void k_fit_total_partial(long long count, const double* __restrict__ v_in, double* __restrict__ part) {
  __shared__ double lds[256];
  double v[1] = {0.0};
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (long long)gridDim.x * 256) v[0] += v_in[i];
  block_sum<1>(v, lds);
  if (threadIdx.x == 0) part[blockIdx.x] = v[0];
}
__global__ __launch_bounds__(256) void k_fit_total(long long count, const double* __restrict__ v_in, double* __restrict__ out) {
  __shared__ double lds[256];
  double v[1] = {0.0};
  for (long long i = threadIdx.x; i < count; i += 256) v[0] += v_in[i];
  block_sum<1>(v, lds);
  if (threadIdx.x == 0) out[0] = v[0];
}
#endif

// ---- host side of fppara ------------------------------------------------------------------------------------------------
namespace fitpack {

// fpdisc.f for k = 3 (k2 = 5): b[(n - 8) * 5], 0-based row l - k2 for l = k2..nk1
inline void fpdisc(const std::vector<double>& t, int n, std::vector<double>& b) {
  const int k2 = 5, k1 = 4, k = 3, nk1 = n - k1, nrint = nk1 - k;
  const double an = nrint, fac = an / (t[nk1] - t[k1 - 1]);
  b.assign((size_t)std::max(0, n - 2 * k1) * k2, 0.0);
  double h[8];
  for (int l = k2; l <= nk1; ++l) {
    const int lmk = l - k1;
    for (int j = 1; j <= k1; ++j) {
      const int ik = j + k1, lj = l + j, lk = lj - k2;
      h[j - 1] = t[l - 1] - t[lk - 1];
      h[ik - 1] = t[l - 1] - t[lj - 1];
    }
    int lp = lmk;
    for (int j = 1; j <= k2; ++j) {
      int jk = j;
      double prod = h[j - 1];
      for (int i = 1; i <= k; ++i) { jk = jk + 1; prod = prod * h[jk - 1] * fac; }
      const int lk = lp + k1;
      b[(size_t)(lmk - 1) * k2 + (j - 1)] = (t[lk - 1] - t[lp - 1]) / prod;
      lp = lp + 1;
    }
  }
}

// fpknot.f: one more knot, in the interval with the largest residual that still holds samples (1-based bookkeeping kept)
inline void fpknot(const double* x, std::vector<double>& t, int& n, std::vector<double>& fpint, std::vector<int>& nrdata, int& nrint) {
  const int k = (n - nrint - 1) / 2;
  double fpmax = 0.0;
  int jbegin = 1, number = 0, maxpt = 0, maxbeg = 0;
  for (int j = 1; j <= nrint; ++j) {
    const int jpoint = nrdata[j - 1];
    if (!(fpmax >= fpint[j - 1] || jpoint == 0)) { fpmax = fpint[j - 1]; number = j; maxpt = jpoint; maxbeg = jbegin; }
    jbegin = jbegin + jpoint + 1;
  }
  const int ihalf = maxpt / 2 + 1, nrx = maxbeg + ihalf, next = number + 1;
  if (next <= nrint) {
    for (int j = next; j <= nrint; ++j) {
      const int jj = next + nrint - j;
      fpint[jj] = fpint[jj - 1];
      nrdata[jj] = nrdata[jj - 1];
      const int jk = jj + k;
      t[jk] = t[jk - 1];
    }
  }
  nrdata[number - 1] = ihalf - 1;
  nrdata[next - 1] = maxpt - ihalf;
  const double am = maxpt;
  double an = nrdata[number - 1];
  fpint[number - 1] = fpmax * an / am;
  an = nrdata[next - 1];
  fpint[next - 1] = fpmax * an / am;
  const int jk = next + k;
  t[jk - 1] = x[nrx - 1];
  n = n + 1;
  nrint = nrint + 1;
}

// `nplus` calls of fpknot (stopping like fppara when n reaches nmax or nest) with the same result in O(nrint + nplus log nrint):
// fpknot scans all intervals for the largest residual and shifts three arrays per knot -- O(n) each, 0.2 s of host time in a
// fit that ends with 25 000 knots.  Here the intervals are a linked list in time order and the candidates (intervals that
// still hold samples) a heap ordered like fpknot's scan: the largest fpint, the EARLIEST interval among equal ones (its scan
// replaces the maximum only by a strictly larger value).  In a state fpknot itself does not handle (no interval with a positive
// residual and samples left: it would index interval 0) the insertions stop.
inline void fpknot_batch(const double* x, std::vector<double>& t, int& n, std::vector<double>& fpint, std::vector<int>& nrdata, int& nrint, int nplus,
                         int nmax, int nest) {
  if (nplus <= 0) return;
  if (nplus < 8 || nrint < 64) {                     // few insertions: the plain calls
    for (int l = 0; l < nplus; ++l) { fpknot(x, t, n, fpint, nrdata, nrint); if (n == nmax || n == nest) break; }
    return;
  }
  const int k = (n - nrint - 1) / 2;
  struct Node { double fp, tright; int nr, jbegin, next, version; };
  std::vector<Node> nodes;
  nodes.reserve((size_t)nrint + (size_t)nplus);
  {
    int jbegin = 1;
    for (int j = 0; j < nrint; ++j) {
      nodes.push_back(Node{fpint[j], j + 1 < nrint ? t[k + 1 + j] : 0.0, nrdata[j], jbegin, j + 1 < nrint ? j + 1 : -1, 0});
      jbegin = jbegin + nrdata[j] + 1;
    }
  }
  struct Cand { double fp; int jbegin, node, version; };
  auto worse = [](const Cand& a, const Cand& b) { return a.fp < b.fp || (a.fp == b.fp && a.jbegin > b.jbegin); };      // heap top = max fp, then min jbegin
  std::vector<Cand> heap;
  heap.reserve((size_t)nrint + 2 * (size_t)nplus);
  for (int j = 0; j < nrint; ++j) if (nodes[j].nr != 0 && nodes[j].fp > 0.0) heap.push_back(Cand{nodes[j].fp, nodes[j].jbegin, j, 0});
  std::make_heap(heap.begin(), heap.end(), worse);
  int added = 0;
  bool fallback = false;
  for (int l = 0; l < nplus; ++l) {
    while (!heap.empty() && heap.front().version != nodes[heap.front().node].version) { std::pop_heap(heap.begin(), heap.end(), worse); heap.pop_back(); }
    if (heap.empty()) { fallback = true; break; }
    std::pop_heap(heap.begin(), heap.end(), worse);
    const Cand c = heap.back();
    heap.pop_back();
    Node& a = nodes[c.node];
    const double fpmax = a.fp;
    const int maxpt = a.nr, maxbeg = a.jbegin;
    const int ihalf = maxpt / 2 + 1, nrx = maxbeg + ihalf;
    Node b;
    b.tright = a.tright; b.next = a.next; b.version = 0;
    a.nr = ihalf - 1;
    b.nr = maxpt - ihalf;
    const double am = maxpt;
    double an = a.nr;
    a.fp = fpmax * an / am;
    an = b.nr;
    b.fp = fpmax * an / am;
    a.tright = x[nrx - 1];
    b.jbegin = nrx;                                   // = maxbeg + (ihalf - 1) + 1
    ++a.version;
    const int bi = (int)nodes.size();
    a.next = bi;
    const Cand ca{a.fp, a.jbegin, c.node, a.version}, cb{b.fp, b.jbegin, bi, 0};
    const bool push_a = a.nr != 0 && a.fp > 0.0, push_b = b.nr != 0 && b.fp > 0.0;
    nodes.push_back(b);                               // (invalidates a)
    if (push_a) { heap.push_back(ca); std::push_heap(heap.begin(), heap.end(), worse); }
    if (push_b) { heap.push_back(cb); std::push_heap(heap.begin(), heap.end(), worse); }
    ++added;
    if (n + added == nmax || n + added == nest) break;
  }
  // back to fpknot's arrays, in time order
  {
    const int nr2 = nrint + added;
    int j = 0;
    for (int i = 0; i != -1; i = nodes[i].next, ++j) {
      fpint[j] = nodes[i].fp;
      nrdata[j] = nodes[i].nr;
      if (nodes[i].next != -1) t[k + 1 + j] = nodes[i].tright;
    }
    // (like fpknot, nothing behind the interior knots is maintained: fppara rewrites the k + 1 end knots before every pass)
    n += added;
    nrint = nr2;
  }
  (void)fallback;
}

inline double fprati(double& p1, double& f1, double p2, double f2, double& p3, double& f3) {
  double p;
  if (p3 > 0.0) {
    const double h1 = f1 * (f2 - f3), h2 = f2 * (f3 - f1), h3 = f3 * (f1 - f2);
    p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3);
  } else {
    p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3);
  }
  if (f2 < 0.0) { p3 = p2; f3 = f2; } else { p1 = p2; f1 = f2; }
  return p;
}

}  // namespace fitpack
}  // namespace mvus
