// Host-side preparation of one BA problem: validation of the mvus_problem description and the
// derived tables the kernels index with (launch chunks, control-point -> x index maps, the
// parameter-independent motion-regulariser samples).  Pure host C++ (no HIP), shared by the HIP
// backend (which uploads these arrays once) and by the test-only host backend.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mvus_ba.h"
#include "ba_math.h"

namespace mvus {

constexpr int kChunk = 256;  // observations per workgroup; a chunk never straddles two cameras

struct HostProblem {
  int C = 0, P = 6, NS = 21, S = 0;
  int calib = 0, undist = 1, rs_free = 0, rs_bounds = 0, motion_reg = 0, motion_type = 0, sync_free = 1;
  double w = 1.0;
  int64_t M = 0, n = 0, m = 0;
  int T = 0, N = 0;
  std::vector<int64_t> det_off;
  std::vector<double> frame, u_raw, v_raw, H, K, dist, istart, iend, knots;
  std::vector<int32_t> knot_off, ctrl_off, xoff, ctrl_x0, ctrl_stride;
  std::vector<int32_t> lut, lut_off;
  std::vector<double> lut_scale;
  std::vector<int32_t> chunk_cam, chunk_count;
  std::vector<int64_t> chunk_start;
  std::vector<ChunkInfo> chunks;
  std::vector<SplineInfo> sinfo;
  std::vector<double> ms_t, ms_basis;
  std::vector<int32_t> ms_ctrl, ms_part, ms_pat, ms_row_lo, ms_row_hi;
  std::vector<int32_t> cam_chunk_off;
  // window-major assembly: time range of the detections whose first control point is >= p (win_tlo[p]) / < p (win_thi[p]),
  // p = 0 .. N; per-camera frame grids (CamWin); frames_sorted: every camera's frames are non-decreasing
  std::vector<double> win_tlo, win_thi;
  std::vector<CamWin> cam_win;
  int64_t flut_len = 0;
  bool frames_sorted = true;

  SplineView spline_view() const {
    return SplineView{S, istart.data(), iend.data(), knots.data(), knot_off.data(), ctrl_off.data(), xoff.data(),
                      lut.data(), lut_off.data(), lut_scale.data(), sinfo.data()};
  }
  MotionView motion_view() const {
    return MotionView{T, motion_type, w, ms_t.data(), ms_basis.data(), ms_ctrl.data(), ms_part.data(), ms_pat.data(),
                      ctrl_x0.data(), ctrl_stride.data(), ms_row_lo.data(), ms_row_hi.data()};
  }

  // launch chunks from det_off (again after detections were removed)
  void build_chunks() {
    chunk_cam.clear(); chunk_start.clear(); chunk_count.clear(); chunks.clear();
    cam_chunk_off.assign(C + 1, 0);
    for (int c = 0; c < C; ++c) {
      cam_chunk_off[c] = (int32_t)chunks.size();
      for (int64_t a = det_off[c]; a < det_off[c + 1]; a += kChunk) {
        const int32_t cnt = (int32_t)std::min<int64_t>(kChunk, det_off[c + 1] - a);
        chunk_cam.push_back(c); chunk_start.push_back(a); chunk_count.push_back(cnt);
        chunks.push_back(ChunkInfo{(long long)a, (long long)det_off[c], (long long)(det_off[c + 1] - det_off[c]), c, cnt});
      }
    }
    cam_chunk_off[C] = (int32_t)chunks.size();
  }

  // Tables of the window-major assembly.  A visible time stamp tau of interval s lies in knot span l (t[l] <= tau < t[l+1], clamped to
  // [3, n_s - 1]) and its first control point is g = ctrl_off[s] + l - 3, non-decreasing in tau.  win_tlo[p] <= every visible tau with
  // g(tau) >= p, win_thi[p] >= every visible tau with g(tau) < p (suffix minimum / prefix maximum of the spans' ends, so the bounds hold
  // for any knot vector the checks above let through).
  void build_window_tables() {
    const double inf = INFINITY;
    std::vector<double> lo(N + 1, inf), hi(N + 1, -inf);
    for (int s = 0; s < S; ++s) {
      const double* t = knots.data() + knot_off[s];
      const int ns = ctrl_off[s + 1] - ctrl_off[s];
      for (int j = 0; j + 4 <= ns; ++j) {                  // span l = j + 3 of interval s
        const int p = ctrl_off[s] + j;
        lo[p] = j == 0 ? istart[s] : std::max(t[j + 3], istart[s]);
        hi[p] = j + 4 == ns ? iend[s] : std::min(t[j + 4], iend[s]);
      }
    }
    win_tlo.assign(N + 1, inf); win_thi.assign(N + 1, -inf);
    double run = inf;
    for (int p = N; p >= 0; --p) { run = std::min(run, lo[p]); win_tlo[p] = run; }
    run = -inf;
    for (int p = 0; p <= N; ++p) { win_thi[p] = run; if (p < N) run = std::max(run, hi[p]); }
    cam_win.assign(C, CamWin{0.0, 1.0, 0.0, 0.0, 0, 1, 0, 0});
    frames_sorted = true;
    int64_t off = 0;
    for (int c = 0; c < C; ++c) {
      const int64_t a = det_off[c], b = det_off[c + 1];
      CamWin& w = cam_win[c];
      w.lut_off = (int32_t)off;
      w.ncell = (int32_t)std::max<int64_t>(1, std::min<int64_t>(b - a, 0x3fffffff));
      if (b > a) {
        double fmin = frame[a], fmax = frame[a], vmin = v_raw[a], vmax = v_raw[a];
        for (int64_t i = a; i < b; ++i) {
          if (i > a && frame[i] < frame[i - 1]) frames_sorted = false;
          fmin = std::min(fmin, frame[i]); fmax = std::max(fmax, frame[i]);
          vmin = std::min(vmin, v_raw[i]); vmax = std::max(vmax, v_raw[i]);
        }
        if (!(fmin == fmin && fmax == fmax && vmin == vmin && vmax == vmax) || std::isinf(fmin) || std::isinf(fmax)) frames_sorted = false;   // NaN / inf: the ordered search is off
        w.f0 = fmin; w.vmin = vmin; w.vmax = vmax;
        w.scale = fmax > fmin ? (double)w.ncell / (fmax - fmin) : 1.0;
      }
      off += (int64_t)w.ncell + 1;
    }
    flut_len = off;
    if (off > 0x7fffffff) frames_sorted = false;
  }

  // returns "" on success, otherwise what is wrong with the description
  std::string build(const mvus_problem* p) {
    if (!p) return "problem is NULL";
    C = p->num_cam;
    if (C < 1) return "num_cam must be >= 1";
    calib = p->opt_calib != 0; undist = p->undist_points != 0; rs_free = p->rs_free != 0;
    rs_bounds = p->rs_bounds != 0; motion_reg = p->motion_reg != 0; motion_type = p->motion_type; sync_free = p->opt_sync != 0;
    if (motion_reg && motion_type != MVUS_MOTION_F && motion_type != MVUS_MOTION_KE)
      return "Motion type must be either F or KE";          // common.py:416
    w = p->motion_weight;
    P = calib ? 15 : 6;
    NS = 3 + P + 12;
    if (!p->det_offsets || !p->img_height || !p->interval || !p->knot_offsets || !p->knots) return "NULL array in problem";
    det_off.assign(p->det_offsets, p->det_offsets + C + 1);
    if (det_off[0] != 0) return "det_offsets[0] must be 0";
    for (int c = 0; c < C; ++c) if (det_off[c + 1] < det_off[c]) return "det_offsets must be non-decreasing";
    M = det_off[C];
    if (M > 0 && (!p->frame || !p->u_raw || !p->v_raw)) return "NULL detection array";
    frame.assign(p->frame, p->frame + M); u_raw.assign(p->u_raw, p->u_raw + M); v_raw.assign(p->v_raw, p->v_raw + M);
    H.assign(p->img_height, p->img_height + C);
    K.assign(4 * C, 0.0); dist.assign(5 * C, 0.0);
    if (!calib) {
      if (!p->K || !p->dist) return "K and dist are required when opt_calib is off";
      K.assign(p->K, p->K + 4 * C); dist.assign(p->dist, p->dist + 5 * C);
    } else if (p->K && p->dist) {
      K.assign(p->K, p->K + 4 * C); dist.assign(p->dist, p->dist + 5 * C);
    }
    S = p->num_splines;
    if (S < 1) return "num_splines must be >= 1";
    istart.assign(p->interval, p->interval + S);
    iend.assign(p->interval + S, p->interval + 2 * S);
    for (int s = 0; s < S; ++s) {
      if (!(iend[s] > istart[s])) return "spline interval with end <= start";
      if (s > 0 && !(istart[s] > iend[s - 1])) return "spline intervals must be sorted and disjoint";   // util.py:82
    }
    knot_off.resize(S + 1); ctrl_off.resize(S + 1); xoff.resize(S);
    ctrl_off[0] = 0;
    if (p->knot_offsets[0] != 0) return "knot_offsets[0] must be 0";
    for (int s = 0; s < S; ++s)
      if (p->knot_offsets[s + 1] < p->knot_offsets[s] + 8 || p->knot_offsets[s + 1] > 0x7fffffff) return "a cubic spline needs at least 8 knots (knot_offsets must increase)";
    for (int s = 0; s <= S; ++s) knot_off[s] = (int32_t)p->knot_offsets[s];
    knots.assign(p->knots, p->knots + knot_off[S]);
    int64_t xo = (int64_t)C * (3 + P);
    for (int s = 0; s < S; ++s) {
      const int nk = knot_off[s + 1] - knot_off[s];
      if (nk < 8) return "a cubic spline needs at least 8 knots";
      const int ns = nk - 4;
      const double* t = knots.data() + knot_off[s];
      for (int k = 1; k < nk; ++k) if (t[k] < t[k - 1]) return "knot vector must be non-decreasing";
      for (int k = 3; k < ns; ++k) if (!(t[k + 1] > t[k])) return "interior knots must be distinct (a spline piece with repeated interior knots -- e.g. the cubic twin of the reference's k=1 fallback for a three-sample trajectory part -- can be evaluated, but not bundle-adjusted: the reference's jac_BA has no pattern for it either)";
      ctrl_off[s + 1] = ctrl_off[s] + ns;
      xoff[s] = (int32_t)xo;
      xo += 3 * (int64_t)ns;
    }
    N = ctrl_off[S];
    n = xo;
    ctrl_x0.resize(N); ctrl_stride.resize(N);
    for (int s = 0; s < S; ++s)
      for (int g = ctrl_off[s]; g < ctrl_off[s + 1]; ++g) { ctrl_x0[g] = xoff[s] + (g - ctrl_off[s]); ctrl_stride[g] = ctrl_off[s + 1] - ctrl_off[s]; }
    // span look-up tables: 2 cells per control point
    lut.clear(); lut_off.assign(S + 1, 0); lut_scale.assign(S, 0.0);
    for (int s = 0; s < S; ++s) {
      const double* t = knots.data() + knot_off[s];
      const int ns = ctrl_off[s + 1] - ctrl_off[s];
      const int nb = 2 * ns;
      const double t0 = t[3], t1 = t[ns];
      lut_scale[s] = nb / (t1 - t0);
      for (int b = 0; b < nb; ++b) lut.push_back(find_span(t, ns, t0 + b / lut_scale[s]));
      lut_off[s + 1] = (int32_t)lut.size();
    }
    sinfo.resize(S);
    for (int s = 0; s < S; ++s) {
      const double* t = knots.data() + knot_off[s];
      sinfo[s] = SplineInfo{istart[s], iend[s], t[3], lut_scale[s], knot_off[s], ctrl_off[s], ctrl_off[s + 1] - ctrl_off[s], xoff[s],
                            lut_off[s], lut_off[s + 1] - lut_off[s], 0, 0};
    }
    build_chunks();
    build_window_tables();
    // motion samples: ts = arange(int[0,0], int[1,-1], 1) kept where start <= ts <= end (common.py:289-292)
    ms_t.clear(); ms_basis.clear(); ms_ctrl.clear(); ms_part.clear(); ms_pat.clear();
    T = 0;
    if (motion_reg) {
      const double a0 = istart[0], b0 = iend[S - 1];
      const int64_t len = (b0 > a0) ? (int64_t)std::ceil(b0 - a0) : 0;
      for (int s = 0; s < S; ++s) {
        const double* t = knots.data() + knot_off[s];
        const int ns = ctrl_off[s + 1] - ctrl_off[s];
        for (int64_t k = 0; k < len; ++k) {
          const double ts = a0 + (double)k;
          if (!(ts >= istart[s] && ts <= iend[s])) continue;
          const int l = find_span(t, ns, ts);
          double h[4], dh[4];
          bspline_basis<false>(t, l, ts, h, dh);
          ms_t.push_back(ts);
          for (int q = 0; q < 4; ++q) ms_basis.push_back(h[q]);
          ms_ctrl.push_back(ctrl_off[s] + l - 3);
          const int part = find_interval(istart.data(), iend.data(), S, ts);     // half-open: util.py:105
          ms_part.push_back(part);
          // reference pattern of the row (common.py:573-585); a sample that is a member of no interval (it sits on a
          // closed end) indexes tck[-1] there: the LAST spline's knots.  Such rows are identically zero, their
          // pattern only matters to the column grouping of the finite differences.
          if (part >= 0) ms_pat.push_back(pattern_canonical(t, ns, ctrl_off[s], l, ts));
          else {
            const double* tl = knots.data() + knot_off[S - 1];
            const int nl = ctrl_off[S] - ctrl_off[S - 1];
            ms_pat.push_back(pattern_canonical(tl, nl, ctrl_off[S - 1], find_span(tl, nl, ts), ts));
          }
        }
      }
      T = (int)ms_t.size();
    }
    // motion rows that can touch control point g (row j uses the samples j-1, j, j+1, each with four control points):
    // a conservative contiguous range, the kernels test the exact membership
    ms_row_lo.assign(N, 0); ms_row_hi.assign(N, 0);
    if (T > 0) {
      std::vector<int32_t> lo(N, T), hi(N, 0);
      for (int j = 0; j < T; ++j)
        for (int k = -1; k <= 1; ++k) {
          const int js = j + k;
          if (js < 0 || js >= T) continue;
          for (int q = -1; q < 5; ++q) {                 // one point of margin: a finite-difference row stores the four
            const int g = ms_ctrl[js] + q;               // points around its PATTERN, which can reach one past the active four
            if (g < 0 || g >= N) continue;
            lo[g] = std::min(lo[g], j); hi[g] = std::max(hi[g], j + 1);
          }
        }
      for (int g = 0; g < N; ++g) { ms_row_lo[g] = lo[g] < hi[g] ? lo[g] : 0; ms_row_hi[g] = lo[g] < hi[g] ? hi[g] : 0; }
    }
    m = 2 * M + T;
    return "";
  }
};

}  // namespace mvus
