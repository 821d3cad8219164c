// Per-observation arithmetic of the bundle-adjustment hot path (fp64).
//
// Everything here is `MVUS_HD` (host+device) so that the HIP kernels in ba_kernels.hip and the
// test-only host harness in tests/hostcheck/ compile the *same* source.  The product library
// (libmvusba.so) only ever calls these from device code.
//
// Reference semantics (CenekAlbl/mvus, multiviewunsynch/reconstruction/common.py):
//   timestamp            common.py:125     tau = alpha*(frame + rs*v_raw/H) + beta
//   visibility           tools/util.py:105 start <= tau < end  (half-open), else residual 0
//   spline point         common.py:331     scipy splev == FITPACK fpbspl de Boor-Cox recurrence
//   rotation             common.py:1136    cv2.Rodrigues(rvec)
//   projection           common.py:1072    x = K [R t] X ; x /= x[2]
//   observed pixel       common.py:1147    K * undistortPoints(raw; K, d)  (5 fixed-point iterations)
//   residual             common.py:357     |x_cal - x_obs| per axis
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MVUS_HD __host__ __device__ __forceinline__
#else
#define MVUS_HD inline
#endif
// Scheduling fence between the groups of Jacobian slots (device code, opt-in): the values of one group are stored before
// the next group is computed, instead of the whole 2 x NS block sitting in registers until one burst of stores at the end.
#if defined(__HIP_DEVICE_COMPILE__) && defined(MVUS_JAC_GROUP_FENCE)
#define MVUS_GROUP_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MVUS_GROUP_FENCE() ((void)0)
#endif

namespace mvus {

// Jacobian slot layout of one residual row (NS = 3 + P + 12 slots):
//   [0] alpha  [1] beta  [2] rs  [3 .. 3+P) camera parameters in the reference's order
//   (P=6: rvec,t ; P=15: fx,fy,cx,cy,rvec,t,k1,k2,p1,p2,k3 -- common.py:1113-1124)
//   [3+P + 3*q + d] control point q (0..3, first = span-3) coordinate d (x,y,z)
constexpr int kSyncSlots = 3;
constexpr int kSplineSlots = 12;
MVUS_HD int num_cam_params(bool calib) { return calib ? 15 : 6; }
MVUS_HD int num_slots(bool calib) { return kSyncSlots + num_cam_params(calib) + kSplineSlots; }

struct CamState {
  double alpha, beta, rs, H;
  double fx, fy, cx, cy;
  double d[5];
  double R[9];  // row-major
  double t[3];
  double W[9];  // R * A(rvec): d(R X)/d rvec = -[R X]x W   (Gallego & Yezzi 2015, eq. 8)
};

// cv2.Rodrigues vector->matrix plus the matrix W used for the rotation-vector derivative.
MVUS_HD void rodrigues(const double r[3], double R[9], double W[9]) {
  const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  const double th = sqrt(th2);
  if (th < 2.220446049250313e-16) {
    for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0) ? 1.0 : 0.0; W[i] = R[i]; }
    return;
  }
  const double c = cos(th), s = sin(th), c1 = 1.0 - c;
  const double kx = r[0] / th, ky = r[1] / th, kz = r[2] / th;
  R[0] = c + c1 * kx * kx;      R[1] = c1 * kx * ky - s * kz; R[2] = c1 * kx * kz + s * ky;
  R[3] = c1 * ky * kx + s * kz; R[4] = c + c1 * ky * ky;      R[5] = c1 * ky * kz - s * kx;
  R[6] = c1 * kz * kx - s * ky; R[7] = c1 * kz * ky + s * kx; R[8] = c + c1 * kz * kz;
  // A = (v v^T + (R^T - I) [v]x) / |v|^2 ;  W = R A
  double Rt_I[9] = {R[0] - 1.0, R[3], R[6], R[1], R[4] - 1.0, R[7], R[2], R[5], R[8] - 1.0};
  const double vx[9] = {0.0, -r[2], r[1], r[2], 0.0, -r[0], -r[1], r[0], 0.0};
  double A[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = r[i] * r[j];
      for (int k = 0; k < 3; ++k) acc += Rt_I[3 * i + k] * vx[3 * k + j];
      A[3 * i + j] = acc / th2;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) acc += R[3 * i + k] * A[3 * k + j];
      W[3 * i + j] = acc;
    }
}

// Decode one camera's parameters out of the BA vector x (reference layout, common.py:454-460).
//   x = [alpha(C) beta(C) rs(C) cam_0(P) ... cam_{C-1}(P) spline...]
MVUS_HD void load_cam_state(const double* x, int C, int c, bool calib, const double* Kfix, const double* dfix,
                            double H, CamState& s) {
  s.alpha = x[c]; s.beta = x[C + c]; s.rs = x[2 * C + c]; s.H = H;
  const int P = num_cam_params(calib);
  const double* v = x + 3 * C + c * P;
  double rv[3];
  if (calib) {
    s.fx = v[0]; s.fy = v[1]; s.cx = v[2]; s.cy = v[3];
    rv[0] = v[4]; rv[1] = v[5]; rv[2] = v[6];
    s.t[0] = v[7]; s.t[1] = v[8]; s.t[2] = v[9];
    for (int k = 0; k < 5; ++k) s.d[k] = v[10 + k];
  } else {
    s.fx = Kfix[4 * c + 0]; s.fy = Kfix[4 * c + 1]; s.cx = Kfix[4 * c + 2]; s.cy = Kfix[4 * c + 3];
    rv[0] = v[0]; rv[1] = v[1]; rv[2] = v[2];
    s.t[0] = v[3]; s.t[1] = v[4]; s.t[2] = v[5];
    for (int k = 0; k < 5; ++k) s.d[k] = dfix[5 * c + k];
  }
  rodrigues(rv, s.R, s.W);
}

// tools/util.py:90-116 (belong=True): index s of the interval with start <= tau < end, or -1.
// Intervals are sorted and disjoint (asserted by find_intervals, util.py:82).
MVUS_HD int find_interval(const double* istart, const double* iend, int S, double tau) {
  int lo = 0, hi = S;  // last interval whose start <= tau
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tau - istart[mid] >= 0.0) lo = mid; else hi = mid;
  }
  const bool in = (tau - istart[lo] >= 0.0) != (tau - iend[lo] >= 0.0);
  return in ? lo : -1;
}

// FITPACK splev.f span search: l in [3, n-1] with t[l] <= x < t[l+1] (clamped), n = #coefficients.
MVUS_HD int find_span(const double* t, int n, double x) {
  int lo = 3, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

// FITPACK fpbspl.f for k=3: the four cubic B-splines that are non-zero on span l, and (optionally)
// their derivatives  B'_i = 3 [ B_{i,2}/(t_{i+3}-t_i) - B_{i+1,2}/(t_{i+4}-t_{i+1}) ].
// The knots enter as the window tt[j] = t[l-2+j], j = 0..5 (all the recurrence touches).
template <bool DERIV>
MVUS_HD void bspline_basis_w(const double tt[6], double x, double h[4], double dh[4]) {
  double hh[3];
  h[0] = 1.0;
  double q[3] = {0.0, 0.0, 0.0};  // quadratic basis B_{l-2,2}, B_{l-1,2}, B_{l,2}
  for (int j = 1; j <= 3; ++j) {
    for (int i = 0; i < j; ++i) hh[i] = h[i];
    h[0] = 0.0;
    for (int i = 0; i < j; ++i) {
      const double tli = tt[i + 3], tlj = tt[i + 3 - j];          // t[l+i+1], t[l+i+1-j]
      const double f = hh[i] / (tli - tlj);
      h[i] = h[i] + f * (tli - x);
      h[i + 1] = f * (x - tlj);
    }
    if (DERIV && j == 2) { q[0] = h[0]; q[1] = h[1]; q[2] = h[2]; }
  }
  if (DERIV) {
    // coefficient i = l-3+k ; quadratic index m = l-2+k' ; denominators t[m+3]-t[m]
    const double e0 = 3.0 * q[0] / (tt[3] - tt[0]);
    const double e1 = 3.0 * q[1] / (tt[4] - tt[1]);
    const double e2 = 3.0 * q[2] / (tt[5] - tt[2]);
    dh[0] = -e0; dh[1] = e0 - e1; dh[2] = e1 - e2; dh[3] = e2;
  }
}
template <bool DERIV>
MVUS_HD void bspline_basis(const double* t, int l, double x, double h[4], double dh[4]) {
  double tt[6];
  for (int j = 0; j < 6; ++j) tt[j] = t[l - 2 + j];
  bspline_basis_w<DERIV>(tt, x, h, dh);
}

// cv2.undistortPoints (5 iterations) on normalised coordinates; with TANGENT also the derivatives of
// the *unrolled* iteration w.r.t. (x0, y0, k1, k2, p1, p2, k3):  dx[7], dy[7].
template <bool TANGENT>
MVUS_HD void undistort5(double x0, double y0, const double d[5], double& xo, double& yo, double dx[7], double dy[7]) {
  const double k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3], k3 = d[4];
  double x = x0, y = y0;
  if (TANGENT) {
    for (int k = 0; k < 7; ++k) { dx[k] = 0.0; dy[k] = 0.0; }
    dx[0] = 1.0; dy[1] = 1.0;
  }
  bool stopped = false;        // OpenCV >= 4.1.1 (cvUndistortPointsInternal, regression_14583): a negative 1 / (1 + k1 r^2 + ...) ends
                               // the iteration with the point reset to its start (x0, y0) -- restated here, in the oracle and in the shim
  for (int it = 0; it < 5; ++it) {
    const double r2 = x * x + y * y;
    const double qd = 1.0 + ((k3 * r2 + k2) * r2 + k1) * r2;
    const double icd = 1.0 / qd;
    if (!stopped && icd < 0.0) {
      stopped = true;
      x = x0; y = y0;
      if (TANGENT) { for (int k = 0; k < 7; ++k) { dx[k] = 0.0; dy[k] = 0.0; } dx[0] = 1.0; dy[1] = 1.0; }
    }
    if (stopped) continue;
    const double ddx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
    const double ddy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
    const double nx = x0 - ddx, ny = y0 - ddy;
    if (TANGENT) {
      const double q_r2 = (3.0 * k3 * r2 + 2.0 * k2) * r2 + k1;
      const double ax = 2.0 * p1 * y + 6.0 * p2 * x, ay = 2.0 * p1 * x + 2.0 * p2 * y;   // d ddx / d(x,y)
      const double bx = 2.0 * p1 * x + 2.0 * p2 * y, by = 6.0 * p1 * y + 2.0 * p2 * x;   // d ddy / d(x,y)
      const double icd2 = icd * icd;
      for (int k = 0; k < 7; ++k) {
        const double tx = dx[k], ty = dy[k];
        const double r2d = 2.0 * (x * tx + y * ty);
        double qdot = q_r2 * r2d;
        double ddxd = ax * tx + ay * ty;
        double ddyd = bx * tx + by * ty;
        if (k == 2) qdot += r2;
        if (k == 3) qdot += r2 * r2;
        if (k == 6) qdot += r2 * r2 * r2;
        if (k == 4) { ddxd += 2.0 * x * y; ddyd += r2 + 2.0 * y * y; }
        if (k == 5) { ddxd += r2 + 2.0 * x * x; ddyd += 2.0 * x * y; }
        const double x0d = (k == 0) ? 1.0 : 0.0, y0d = (k == 1) ? 1.0 : 0.0;
        dx[k] = (x0d - ddxd) * icd - nx * icd2 * qdot;
        dy[k] = (y0d - ddyd) * icd - ny * icd2 * qdot;
      }
    }
    x = nx * icd;
    y = ny * icd;
  }
  xo = x; yo = y;
}

// One launch chunk (<= 256 consecutive detections of one camera), 32 bytes: one scalar load per workgroup.
struct ChunkInfo {
  long long start;          // first detection (camera-segmented index)
  long long cam_start;      // det_offsets[cam]
  long long cam_count;      // detections of the camera (M_c): the y residuals sit M_c after the x residuals
  int32_t cam, count;
};

// Everything the kernels need to know about one spline interval, in one 64-byte record: after the interval of a
// timestamp is known a single (vector) load fetches it, instead of a chain of dependent table reads.
struct SplineInfo {
  double istart, iend;      // spline['int'][:, s]
  double t3;                // first knot of the interval (origin of the span look-up grid)
  double lut_scale;         // cells per unit time
  int32_t knot_off;         // offset of the knot vector in `knots`
  int32_t ctrl_off;         // global index of the first control point
  int32_t n;                // number of coefficients
  int32_t xoff;             // index in x of the coefficient block cx(n) cy(n) cz(n)
  int32_t lut_off, nb;      // span look-up table: offset and number of cells
  int32_t pad0, pad1;
};

// Window-major assembly (ba_assemble_win.hip.h): where, among the frame-sorted detections of one camera, a time range can lie.
// Cell k of the camera's grid covers frames [f0 + k / scale, f0 + (k + 1) / scale); flut[lut_off + k], k = 0 .. ncell, is the first
// detection (camera-local index) whose frame is >= the left edge of cell k (flut[lut_off + ncell] = M_c).  vmin / vmax bound v_raw over
// the camera's detections, i.e. the rolling-shutter term rs * v / H of the time stamp.  The grid is laid out once per problem
// (removing detections only thins the cells out; the table is rebuilt, the grid stays).
struct CamWin {
  double f0, scale;
  double vmin, vmax;
  int32_t lut_off, ncell;
  int32_t pad0, pad1;
};

// Read-only view of the trajectory splines as they sit in device memory.
struct SplineView {
  int S;                    // number of spline intervals
  const double* istart;     // [S]
  const double* iend;       // [S]
  const double* knots;      // concatenated knot vectors
  const int32_t* knot_off;  // [S+1] offsets into knots
  const int32_t* ctrl_off;  // [S+1] cumulative number of control points (global control index base)
  const int32_t* xoff;      // [S]  index in x of spline s coefficient block: cx(n_s) cy(n_s) cz(n_s)
  // optional span look-up table (nullptr -> binary search): per spline a uniform grid over [start, end] whose
  // cell b stores the span of the cell's left edge, so the search is one table read plus ~1 forward step
  const int32_t* lut;       // concatenated tables
  const int32_t* lut_off;   // [S+1]
  const double* lut_scale;  // [S] cells per unit time
  const SplineInfo* info;   // [S] the same facts, one record per interval (nullptr -> the separate tables above)
};

// Same result as find_span, through the look-up table when the view carries one.
MVUS_HD int find_span_lut(const SplineView& sp, int s, const double* t, int n, double x) {
  if (!sp.lut) return find_span(t, n, x);
  const int nb = sp.lut_off[s + 1] - sp.lut_off[s];
  int b = (int)((x - t[3]) * sp.lut_scale[s]);
  b = b < 0 ? 0 : (b >= nb ? nb - 1 : b);
  int l = sp.lut[sp.lut_off[s] + b];
  while (l > 3 && t[l] > x) --l;                 // rounding of the cell index can land one cell too far
  while (l + 1 < n && t[l + 1] <= x) ++l;
  return l;
}

// Where a timestamp falls: interval, knot span l (FITPACK rule t[l] <= tau < t[l+1], clamped to [3, n-1]), the six knots
// t[l-2 .. l+3] and the twelve active coefficients.  Written for a SHORT chain of dependent memory accesses, which is
// what a wavefront of this kernel waits on: [interval record] -> [table cell] -> [8 knots + 18 coefficients around the
// guessed span, issued together] -> arithmetic; the guess is corrected by +-1 from the loaded window (select, no
// reload) and only a larger miss (strongly non-uniform knots) re-searches.  Results are those of find_interval +
// find_span + plain indexing, bit for bit.
struct SpanLoc {
  int32_t n, xoff, ctrl;    // coefficients of the spline, index of its block in x, global index of control point l-3
  double tt[6];             // t[l-2 .. l+3]
  double c[3][4];           // coefficients l-3 .. l of x, y, z
};
MVUS_HD bool locate_span(const SplineView& sp, const double* x, double tau, SpanLoc& loc) {
  int l;
  const double* t;
  if (sp.info && sp.lut) {
    int s = 0;
    if (sp.S > 1) { s = find_interval(sp.istart, sp.iend, sp.S, tau); if (s < 0) return false; }
    const SplineInfo si = sp.info[s];
    if (sp.S == 1 && ((tau - si.istart >= 0.0) == (tau - si.iend >= 0.0))) return false;    // util.py:105, half-open
    t = sp.knots + si.knot_off;
    int b = (int)((tau - si.t3) * si.lut_scale);
    b = b < 0 ? 0 : (b >= si.nb ? si.nb - 1 : b);
    const int l0 = sp.lut[si.lut_off + b];
    loc.n = si.n; loc.xoff = si.xoff;
    double tk[8], cw[3][6];
    for (int j = 0; j < 8; ++j) tk[j] = t[l0 - 3 + j];                    // t[l0-3 .. l0+4]: always inside the n+4 knots
    const double* cb = x + si.xoff;
    for (int j = 0; j < 6; ++j) {                                          // c[l0-4 .. l0+1], clamped (the clamped ones are never selected)
      int idx = l0 - 4 + j;
      idx = idx < 0 ? 0 : (idx > si.n - 1 ? si.n - 1 : idx);
      cw[0][j] = cb[idx]; cw[1][j] = cb[idx + si.n]; cw[2][j] = cb[idx + 2 * si.n];
    }
    int off = 0;
    bool ok = true;
    if (l0 > 3 && tk[3] > tau) { off = -1; ok = !(l0 - 1 > 3 && tk[2] > tau); }
    else if (l0 + 1 < si.n && tk[4] <= tau) { off = 1; ok = !(l0 + 2 < si.n && tk[5] <= tau); }
    if (ok) {
      l = l0 + off;
      for (int j = 0; j < 6; ++j) loc.tt[j] = off < 0 ? tk[j] : (off == 0 ? tk[j + 1] : tk[j + 2]);
      for (int d = 0; d < 3; ++d)
        for (int q = 0; q < 4; ++q) loc.c[d][q] = off < 0 ? cw[d][q] : (off == 0 ? cw[d][q + 1] : cw[d][q + 2]);
      loc.ctrl = si.ctrl_off + (l - 3);
      return true;
    }
    l = find_span(t, si.n, tau);
    loc.ctrl = si.ctrl_off + (l - 3);
  } else {
    const int s = find_interval(sp.istart, sp.iend, sp.S, tau);
    if (s < 0) return false;
    t = sp.knots + sp.knot_off[s];
    loc.n = sp.ctrl_off[s + 1] - sp.ctrl_off[s];
    loc.xoff = sp.xoff[s];
    l = find_span_lut(sp, s, t, loc.n, tau);
    loc.ctrl = sp.ctrl_off[s] + (l - 3);
  }
  for (int j = 0; j < 6; ++j) loc.tt[j] = t[l - 2 + j];
  const double* cb = x + loc.xoff + (l - 3);
  for (int d = 0; d < 3; ++d)
    for (int q = 0; q < 4; ++q) loc.c[d][q] = cb[q + d * loc.n];
  return true;
}

struct ObsResult {
  double ex, ey;     // |residual| per axis (0 when not visible)
  int32_t ctrl;      // global index of the first of the 4 active control points, -1 when not visible
};

// One observation: residual and (JAC) its 2 x NS Jacobian block, rows pre-multiplied by sign(r) so
// that they are the derivatives of the absolute residuals the reference returns (common.py:357-358).
//   jx/jy : NS values each (see slot layout above); untouched when the observation is not visible.
//   u_obs/v_obs: observed pixel when calibration is fixed (undistorted once at create time).
template <bool CALIB, bool JAC, class Sink>
MVUS_HD ObsResult eval_observation_at(const CamState& cam, double tau, const SpanLoc& loc, bool undist, bool rs_free, bool sync_free,
                                      double frame, double u_raw, double v_raw, double u_obs, double v_obs, Sink& sink);
template <bool CALIB, bool JAC, class Sink>
MVUS_HD ObsResult eval_observation_to(const CamState& cam, const SplineView& sp, const double* x, bool undist, bool rs_free, bool sync_free,
                                      double frame, double u_raw, double v_raw, double u_obs, double v_obs, Sink& sink) {
  const double tau = cam.alpha * (frame + cam.rs * v_raw / cam.H) + cam.beta;
  SpanLoc loc;
  if (!locate_span(sp, x, tau, loc)) { ObsResult out; out.ex = 0.0; out.ey = 0.0; out.ctrl = -1; return out; }
  return eval_observation_at<CALIB, JAC>(cam, tau, loc, undist, rs_free, sync_free, frame, u_raw, v_raw, u_obs, v_obs, sink);
}
// the same observation once its knot span is known (loc: the six knots, the twelve coefficients, the first control point)
template <bool CALIB, bool JAC, class Sink>
MVUS_HD ObsResult eval_observation_at(const CamState& cam, double tau, const SpanLoc& loc, bool undist, bool rs_free, bool sync_free,
                                      double frame, double u_raw, double v_raw, double u_obs, double v_obs, Sink& sink) {
  ObsResult out;
  out.ex = 0.0; out.ey = 0.0; out.ctrl = -1;
  double h[4], dh[4];
  bspline_basis_w<JAC>(loc.tt, tau, h, dh);
  double X[3] = {0.0, 0.0, 0.0}, Xd[3] = {0.0, 0.0, 0.0};
  for (int q = 0; q < 4; ++q) {
    X[0] = X[0] + loc.c[0][q] * h[q]; X[1] = X[1] + loc.c[1][q] * h[q]; X[2] = X[2] + loc.c[2][q] * h[q];
    if (JAC) { Xd[0] += loc.c[0][q] * dh[q]; Xd[1] += loc.c[1][q] * dh[q]; Xd[2] += loc.c[2][q] * dh[q]; }
  }
  const double* R = cam.R;
  const double y0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
  const double y1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
  const double y2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
  const double Xc0 = y0 + cam.t[0], Xc1 = y1 + cam.t[1], Xc2 = y2 + cam.t[2];
  const double iz = 1.0 / Xc2;
  const double xn = Xc0 * iz, yn = Xc1 * iz;
  const double uh = cam.fx * xn + cam.cx;
  const double vh = cam.fy * yn + cam.cy;

  double uo = u_obs, vo = v_obs;
  double oxn = 0.0, oyn = 0.0, ox0 = 0.0, oy0 = 0.0, tdx[7], tdy[7];
  if (CALIB) {
    if (undist) {
      ox0 = (u_raw - cam.cx) / cam.fx;
      oy0 = (v_raw - cam.cy) / cam.fy;
      undistort5<JAC>(ox0, oy0, cam.d, oxn, oyn, tdx, tdy);
      uo = cam.fx * oxn + cam.cx;
      vo = cam.fy * oyn + cam.cy;
    } else {
      uo = u_raw; vo = v_raw;
    }
  }
  const double ru = uh - uo, rv = vh - vo;
  out.ex = fabs(ru); out.ey = fabs(rv);
  out.ctrl = loc.ctrl;
  if (!JAC) return out;
  sink.begin(out.ctrl);

  const double su = (ru < 0.0) ? -1.0 : 1.0, sv = (rv < 0.0) ? -1.0 : 1.0;
  // d(uh)/dXc, d(vh)/dXc, signed
  const double au0 = su * cam.fx * iz, au2 = -su * cam.fx * xn * iz;
  const double av1 = sv * cam.fy * iz, av2 = -sv * cam.fy * yn * iz;
  // g = a * R : derivative w.r.t. the world point X
  const double gu0 = au0 * R[0] + au2 * R[6], gu1 = au0 * R[1] + au2 * R[7], gu2 = au0 * R[2] + au2 * R[8];
  const double gv0 = av1 * R[3] + av2 * R[6], gv1 = av1 * R[4] + av2 * R[7], gv2 = av1 * R[5] + av2 * R[8];
  // time: d r / d tau = g . X'(tau)
  const double dtu = gu0 * Xd[0] + gu1 * Xd[1] + gu2 * Xd[2];
  const double dtv = gv0 * Xd[0] + gv1 * Xd[1] + gv2 * Xd[2];
  const double row = cam.rs * v_raw / cam.H;
  const double dta = frame + row;                       // d tau / d alpha
  const double dtr = rs_free ? cam.alpha * v_raw / cam.H : 0.0;   // d tau / d rs (column absent from the
                                                                  // reference pattern when rs=False, common.py:518-521)
  const double sf = sync_free ? 1.0 : 0.0;               // opt_sync off: alpha, beta leave the pattern (common.py:512-515)
  sink.x(0, sf * dtu * dta); sink.x(1, sf * dtu); sink.x(2, dtu * dtr);
  sink.y(0, sf * dtv * dta); sink.y(1, sf * dtv); sink.y(2, dtv * dtr);
  MVUS_GROUP_FENCE();
  // rotation vector: d Xc / d r = -[y]x W  ->  row a gives (y x a)^T W
  const double cu0 = y1 * au2 - y2 * 0.0, cu1 = y2 * au0 - y0 * au2, cu2 = y0 * 0.0 - y1 * au0;   // y x au, au = (au0,0,au2)
  const double cv0 = y1 * av2 - y2 * av1, cv1 = y2 * 0.0 - y0 * av2, cv2 = y0 * av1 - y1 * 0.0;   // y x av, av = (0,av1,av2)
  const double* W = cam.W;
  const int o = CALIB ? 7 : 3;   // slot of rvec
  sink.x(o + 0, cu0 * W[0] + cu1 * W[3] + cu2 * W[6]);
  sink.x(o + 1, cu0 * W[1] + cu1 * W[4] + cu2 * W[7]);
  sink.x(o + 2, cu0 * W[2] + cu1 * W[5] + cu2 * W[8]);
  sink.y(o + 0, cv0 * W[0] + cv1 * W[3] + cv2 * W[6]);
  sink.y(o + 1, cv0 * W[1] + cv1 * W[4] + cv2 * W[7]);
  sink.y(o + 2, cv0 * W[2] + cv1 * W[5] + cv2 * W[8]);
  MVUS_GROUP_FENCE();
  // translation
  sink.x(o + 3, au0); sink.x(o + 4, 0.0); sink.x(o + 5, au2);
  sink.y(o + 3, 0.0); sink.y(o + 4, av1); sink.y(o + 5, av2);
  MVUS_GROUP_FENCE();
  if (CALIB) {
    // r_u = fx*xn + cx - (fx*oxn + cx),  oxn = undist((u_raw-cx)/fx, (v_raw-cy)/fy; d)
    double duo[9], dvo[9];   // d(uo), d(vo) / d(fx,fy,cx,cy,k1,k2,p1,p2,k3)
    if (undist) {
      duo[0] = oxn - ox0 * tdx[0];
      duo[1] = -cam.fx * tdx[1] * oy0 / cam.fy;
      duo[2] = 1.0 - tdx[0];
      duo[3] = -cam.fx * tdx[1] / cam.fy;
      dvo[0] = -cam.fy * tdy[0] * ox0 / cam.fx;
      dvo[1] = oyn - oy0 * tdy[1];
      dvo[2] = -cam.fy * tdy[0] / cam.fx;
      dvo[3] = 1.0 - tdy[1];
      for (int k = 0; k < 5; ++k) { duo[4 + k] = cam.fx * tdx[2 + k]; dvo[4 + k] = cam.fy * tdy[2 + k]; }
    } else {
      for (int k = 0; k < 9; ++k) { duo[k] = 0.0; dvo[k] = 0.0; }
    }
    sink.x(3, su * (xn - duo[0])); sink.x(4, su * (-duo[1])); sink.x(5, su * (1.0 - duo[2])); sink.x(6, su * (-duo[3]));
    sink.y(3, sv * (-dvo[0])); sink.y(4, sv * (yn - dvo[1])); sink.y(5, sv * (-dvo[2])); sink.y(6, sv * (1.0 - dvo[3]));
    for (int k = 0; k < 5; ++k) { sink.x(13 + k, -su * duo[4 + k]); sink.y(13 + k, -sv * dvo[4 + k]); }
  }
  const int b = kSyncSlots + (CALIB ? 15 : 6);
  if constexpr (Sink::kFactored) {
    // the twelve spline slots are the Kronecker product h (x) [gu; gv]: a sink that forms J^T J itself takes the factors
    // (the window-major assembly: band blocks = (h h^T) (x) (g^T g), 46 multiply-adds per row pair instead of 180)
    sink.factored(h, gu0, gu1, gu2, gv0, gv1, gv2);
  } else {
    for (int q = 0; q < 4; ++q) {
      sink.x(b + 3 * q + 0, h[q] * gu0); sink.x(b + 3 * q + 1, h[q] * gu1); sink.x(b + 3 * q + 2, h[q] * gu2);
      sink.y(b + 3 * q + 0, h[q] * gv0); sink.y(b + 3 * q + 1, h[q] * gv1); sink.y(b + 3 * q + 2, h[q] * gv2);
      MVUS_GROUP_FENCE();
    }
  }
  return out;
}

// Sink that keeps the 2 x NS values in two arrays (the host harness, and kernels that post-process the row).
struct ArraySink {
  static constexpr bool kFactored = false;
  double *jx, *jy;
  MVUS_HD void begin(int32_t) {}
  MVUS_HD void x(int k, double v) { jx[k] = v; }
  MVUS_HD void y(int k, double v) { jy[k] = v; }
};
template <bool CALIB, bool JAC>
MVUS_HD ObsResult eval_observation(const CamState& cam, const SplineView& sp, const double* x, bool undist, bool rs_free, bool sync_free,
                                   double frame, double u_raw, double v_raw, double u_obs, double v_obs,
                                   double* jx, double* jy) {
  ArraySink sink{jx, jy};
  return eval_observation_to<CALIB, JAC>(cam, sp, x, undist, rs_free, sync_free, frame, u_raw, v_raw, u_obs, v_obs, sink);
}

// Reference sparsity pattern for the spline columns of one row (common.py:559-563): the three
// coefficients whose centre knots c_j = t[j+2] (`knot = t[2:-2]`) are nearest to tau.  On FITPACK-like knot
// vectors these are three of the four active coefficients, but nothing in the reference enforces that: with a
// strongly non-uniform knot spacing the window can hold an inactive coefficient, and it does so here too.
// Pattern code of one row: the in-pattern control points as  p | (mask << 25)  with p the global index of the
// lowest one and bit k of the 4-bit mask set when p + k is in the pattern (-1 = all-zero row).  The canonical
// nearest-three rule gives three consecutive points (mask 0b0111).  Two centre knots are repeated in t[2:-2]
// (coefficients 0,1 share the interval start, n-2,n-1 its end); when exactly one of such twins is among the three
// nearest, which one np.argsort returns is decided by numpy's sort kernel (it differs between the scalar, AVX2
// and AVX512 builds), so the reference's pattern is implementation defined in those rows: {0,2,3} (mask 0b1101)
// or {1,2,3}, {n-4,n-3,n-1} (0b1011) or {n-4,n-3,n-2}.  The kernels compute the canonical code (the twin that is
// an ACTIVE coefficient) and flag the row with kPatTie; a caller that holds the reference's own matrix -- it is
// an input of the least_squares call, common.py:670 -- uploads the codes it implies (mvus_ba_upload_pattern).
constexpr int kPatShift = 25;
constexpr int32_t kPatIndexMask = (1 << kPatShift) - 1;
constexpr int32_t kPatTie = 1 << 30;           // output flag of observation_pattern / motion tables only
constexpr int32_t kPatCanon = 0x7;
MVUS_HD int32_t pattern_code(int32_t p, int32_t mask) { return p | (mask << kPatShift); }
MVUS_HD int32_t pattern_index(int32_t code) { return code & kPatIndexMask; }
MVUS_HD int32_t pattern_mask(int32_t code) { return (code >> kPatShift) & 0xf; }
MVUS_HD bool pattern_has(int32_t code, int32_t g) {
  const int32_t d = g - pattern_index(code);
  return d >= 0 && d < 4 && ((pattern_mask(code) >> d) & 1);
}
// canonical code of a row with timestamp tau in span l (t[l] <= tau < t[l+1], i.e. c_{l-2} <= tau < c_{l-1}) of a
// spline with n coefficients starting at global control point c0: walk outwards from tau, taking the nearer of the
// next centre on the left / right three times (exact ties between DISTINCT centres go left, like a stable sort).
// Flagged when the window holds one member of a twin pair without the other.
MVUS_HD int32_t pattern_canonical(const double* t, int n, int c0, int l, double tau) {
  int lo = l - 2, hi = l - 1;          // next candidates
  int first = hi, last = lo;           // chosen window, still empty
  for (int k = 0; k < 3; ++k) {
    const bool hl = lo >= 0, hr = hi <= n - 1;
    const bool left = (hl && hr) ? (fabs(tau - t[lo + 2]) <= fabs(t[hi + 2] - tau)) : hl;
    if (left) { first = lo; --lo; } else { last = hi; ++hi; }
  }
  int32_t code = pattern_code(c0 + first, kPatCanon);
  if (first == 1 || last == n - 2) code |= kPatTie;
  return code;
}

// Pattern of one detection row at x0 (jac_BA, common.py:553-566): code of the three in-pattern control points
// (canonical, see above), or -1 when the detection is not visible at x0 (all-zero row, common.py:566).
MVUS_HD int32_t observation_pattern(const CamState& cam, const SplineView& sp, double frame, double v_raw) {
  const double tau = cam.alpha * (frame + cam.rs * v_raw / cam.H) + cam.beta;
  const int s = find_interval(sp.istart, sp.iend, sp.S, tau);
  if (s < 0) return -1;
  const double* t = sp.knots + sp.knot_off[s];
  const int n = sp.ctrl_off[s + 1] - sp.ctrl_off[s];
  const int l = find_span(t, n, tau);
  return pattern_canonical(t, n, sp.ctrl_off[s], l, tau);
}

// Keep only the spline slots whose control point lies in the pattern.
// base = index of the first spline slot (3 + P); ctrl = first active control point of the row.
MVUS_HD void mask_to_pattern(double* jx, double* jy, int base, int32_t ctrl, int32_t pat) {
  for (int q = 0; q < 4; ++q) {
    if (!pattern_has(pat, ctrl + q))
      for (int d = 0; d < 3; ++d) { jx[base + 3 * q + d] = 0.0; jy[base + 3 * q + d] = 0.0; }
  }
}

// Where the four stored control points of a finite-difference row start: the pattern's points must all be among
// base .. base+3 of ONE spline (ctrl_x0 runs consecutively inside a spline).
MVUS_HD int32_t pattern_fd_base(int32_t code, int N, const int32_t* ctrl_x0) {
  const int32_t p = pattern_index(code);
  const bool room = (p + 3 < N) && (ctrl_x0[p + 3] == ctrl_x0[p] + 3);
  return room ? p : p - 1;            // a canonical triple that ends its spline: shift down by one
}

// ---------------------------------------------------------------------------------------------
// Motion regulariser rows (Scene.error_motion(motion_reg=True) + motion_prior, common.py:362-424,
// 959-1001).  The sample times ts = arange(int[0,0], int[1,-1], 1) kept per interval (closed ends,
// common.py:289-292) do not depend on the parameters, so span, basis values, part membership
// (half-open, util.py:105) and the reference pattern are precomputed once per problem.
// ---------------------------------------------------------------------------------------------
struct MotionView {
  int T;                  // number of samples = number of motion rows
  int type;               // 0 = 'F', 1 = 'KE'
  double w;               // motion_weights
  const double* t;        // [T] sample times
  const double* basis;    // [4*T] cubic basis values at the sample
  const int32_t* ctrl;    // [T] global index of the first active control point
  const int32_t* part;    // [T] interval the sample is a half-open member of, -1 = none (row stays 0)
  const int32_t* pat;     // [T] pattern code of the row (pattern_code; common.py:573-585)
  const int32_t* ctrl_x0;     // [N] x-index of coordinate 0 of control point g
  const int32_t* ctrl_stride; // [N] n_s of the spline control point g belongs to
  const int32_t* row_lo;      // [N] motion rows that can touch control point g: row_lo[g] <= j < row_hi[g]
  const int32_t* row_hi;      // [N]
};

MVUS_HD void motion_point(const MotionView& mv, const double* x, int j, double X[3]) {
  const int g = mv.ctrl[j];
  const int i0 = mv.ctrl_x0[g], st = mv.ctrl_stride[g];
  const double* b = mv.basis + 4 * j;
  for (int d = 0; d < 3; ++d) {
    const double* c = x + i0 + d * st;
    double acc = 0.0;
    for (int q = 0; q < 4; ++q) acc = acc + c[q] * b[q];
    X[d] = acc;
  }
}

// Row j of the motion block, its Jacobian entries handed to `sink(k, q, d, value)` as they are produced (sample k = 0: j-1, 1: j,
// 2: j+1; control point q, coordinate d; entries never handed over are zero).  cidx[3]: first control point of each sample
// (-1 unused).  masked: keep only the reference pattern.
template <bool JAC, class Emit>
MVUS_HD double eval_motion_row_to(const MotionView& mv, const double* x, int j, bool masked, Emit&& emit, int32_t cidx[3]) {
  const double eps = 1e-20;
  if (JAC) { cidx[0] = cidx[1] = cidx[2] = -1; }
  const int p = mv.part[j];
  if (p < 0 || j < 1 || mv.part[j - 1] != p) return 0.0;
  double Xm[3], X0[3];
  motion_point(mv, x, j - 1, Xm);
  motion_point(mv, x, j, X0);
  double row = 0.0;
  double sgn[3];
  const int32_t pc = (JAC && masked) ? mv.pat[j] : 0;
  if (mv.type == 1) {                                    // 'KE'  common.py:976-981
    const double dt = mv.t[j] - mv.t[j - 1];
    double dcoef[3];
    for (int d = 0; d < 3; ++d) {
      const double vel = (X0[d] - Xm[d]) / (dt + eps);
      const double r = mv.w * 0.5 * (vel * vel * dt);
      row += fabs(r);
      sgn[d] = (r < 0.0) ? -1.0 : 1.0;
      dcoef[d] = mv.w * vel * dt / (dt + eps);
    }
    if (JAC) {
      cidx[0] = mv.ctrl[j - 1]; cidx[1] = mv.ctrl[j];
      for (int q = 0; q < 4; ++q) {
        const bool k0 = !masked || pattern_has(pc, cidx[0] + q), k1 = !masked || pattern_has(pc, cidx[1] + q);
        for (int d = 0; d < 3; ++d) {
          if (k0) emit(0, q, d, -sgn[d] * dcoef[d] * mv.basis[4 * (j - 1) + q]);
          if (k1) emit(1, q, d, sgn[d] * dcoef[d] * mv.basis[4 * j + q]);
        }
      }
    }
  } else {                                               // 'F'   common.py:984-998
    if (j + 1 >= mv.T || mv.part[j + 1] != p) return 0.0;
    double Xp[3];
    motion_point(mv, x, j + 1, Xp);
    const double dt1 = mv.t[j] - mv.t[j - 1], dt2 = mv.t[j + 1] - mv.t[j], dt3 = dt1 + dt2;
    for (int d = 0; d < 3; ++d) {
      const double v1 = (X0[d] - Xm[d]) / (dt1 + eps);
      const double v2 = (Xp[d] - X0[d]) / (dt2 + eps);
      const double accel = (v2 - v1) / (dt3 + eps);
      const double r = mv.w * (accel * dt3);
      row += fabs(r);
      sgn[d] = (r < 0.0) ? -1.0 : 1.0;
    }
    if (JAC) {
      const double k3 = mv.w * dt3 / (dt3 + eps);
      double coef[3];
      coef[0] = k3 / (dt1 + eps);
      coef[1] = -k3 * (1.0 / (dt2 + eps) + 1.0 / (dt1 + eps));
      coef[2] = k3 / (dt2 + eps);
      for (int k = 0; k < 3; ++k) {
        cidx[k] = mv.ctrl[j - 1 + k];
        for (int q = 0; q < 4; ++q) {
          if (masked && !pattern_has(pc, cidx[k] + q)) continue;
          for (int d = 0; d < 3; ++d) emit(k, q, d, sgn[d] * coef[k] * mv.basis[4 * (j - 1 + k) + q]);
        }
      }
    }
  }
  return row;
}
// the same with the 36 entries in an array: jrow[12*k + 3*q + d]
template <bool JAC>
MVUS_HD double eval_motion_row(const MotionView& mv, const double* x, int j, bool masked, double* jrow, int32_t cidx[3]) {
  if (JAC) { for (int k = 0; k < 36; ++k) jrow[k] = 0.0; }
  return eval_motion_row_to<JAC>(mv, x, j, masked, [&](int k, int q, int d, double v) { jrow[12 * k + 3 * q + d] = v; }, cidx);
}

// scipy 2-point step for one variable (scipy/optimize/_numdiff.py:146-192, :13-90 with scheme '1-sided', num_steps 1):
// h = sqrt(eps) * sign(x) * max(1, |x|), flipped / shrunk so that x + h stays inside [lb, ub].
MVUS_HD double fd_step(double x, double lb, double ub) {
  const double rstep = 1.4901161193847656e-08;   // sqrt(2.220446049250313e-16)
  double h = rstep * (x >= 0.0 ? 1.0 : -1.0) * fmax(1.0, fabs(x));
  if (lb == -INFINITY && ub == INFINITY) return h;
  const double lower = x - lb, upper = ub - x, xn = x + h;
  const bool violated = (xn < lb) || (xn > ub);
  const bool fitting = fabs(h) <= fmax(lower, upper);
  if (violated && fitting) h = -h;
  if (!fitting) h = (upper >= lower) ? upper : -lower;
  return h;
}

}  // namespace mvus
