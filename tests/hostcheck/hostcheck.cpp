// TEST-ONLY host build of the per-observation device math (mvus_amd/csrc/ba_math.h).
// Lets the CPU test-suite check the arithmetic the HIP kernels run (residual, analytic Jacobian)
// against the oracle without a GPU.  Never linked into libmvusba.so.
#include "../../mvus_amd/csrc/ba_math.h"
#include "../../mvus_amd/csrc/triangulate.hip.h"
#include "../../mvus_amd/csrc/spline_fit.hip.h"
#include "../../mvus_amd/csrc/pnp.hip.h"

using namespace mvus;

extern "C" int hostcheck_eval(int C, int calib, int undist, int rs_free, const int64_t* det_off,
                              const double* frame, const double* u_raw, const double* v_raw, const double* H,
                              const double* Kfix, const double* dfix, int S, const double* istart,
                              const double* iend, const double* knots, const int32_t* knot_off,
                              const int32_t* ctrl_off, const int32_t* xoff, const double* x, double* ex,
                              double* ey, int32_t* ctrl, double* J) {
  SplineView sp{S, istart, iend, knots, knot_off, ctrl_off, xoff};
  const int NS = num_slots(calib != 0);
  for (int c = 0; c < C; ++c) {
    CamState cam;
    load_cam_state(x, C, c, calib != 0, Kfix, dfix, H[c], cam);
    for (int64_t i = det_off[c]; i < det_off[c + 1]; ++i) {
      double uo = u_raw[i], vo = v_raw[i];
      if (!calib && undist) {
        double xn, yn;
        undistort5<false>((u_raw[i] - cam.cx) / cam.fx, (v_raw[i] - cam.cy) / cam.fy, cam.d, xn, yn, nullptr, nullptr);
        uo = cam.fx * xn + cam.cx;
        vo = cam.fy * yn + cam.cy;
      }
      double* jx = J + (2 * i) * NS;
      double* jy = J + (2 * i + 1) * NS;
      for (int k = 0; k < 2 * NS; ++k) jx[k] = 0.0;
      ObsResult r = calib ? eval_observation<true, true>(cam, sp, x, undist != 0, rs_free != 0, true, frame[i], u_raw[i], v_raw[i], uo, vo, jx, jy)
                          : eval_observation<false, true>(cam, sp, x, undist != 0, rs_free != 0, true, frame[i], u_raw[i], v_raw[i], uo, vo, jx, jy);
      ex[i] = r.ex; ey[i] = r.ey; ctrl[i] = r.ctrl;
    }
  }
  return 0;
}

// host build of the per-pair triangulation math (mvus_amd/csrc/triangulate.hip.h); x1, x2: [2][N], X: [4][N]
extern "C" int hostcheck_triangulate(long long N, const double* x1, const double* x2, const double* P1, const double* P2,
                                     double* X, double* err1, double* err2) {
  for (long long i = 0; i < N; ++i) {
    double Xh[4];
    triangulate_pair(P1, P2, x1[i], x1[N + i], x2[i], x2[N + i], Xh);
    for (int k = 0; k < 4; ++k) X[k * N + i] = Xh[k];
    if (err1) err1[i] = reprojection_distance(P1, Xh, x1[i], x1[N + i]);
    if (err2) err2[i] = reprojection_distance(P2, Xh, x2[i], x2[N + i]);
  }
  return 0;
}

// host parts of the smoothing-spline fit (mvus_amd/csrc/spline_fit.hip.h): FITPACK's scalar routines and the double-double
// arithmetic of the ill-conditioned passes
extern "C" void hostcheck_fpdisc(int n, const double* t, double* b_out) {
  std::vector<double> tv(t, t + n), b;
  fitpack::fpdisc(tv, n, b);
  for (size_t i = 0; i < b.size(); ++i) b_out[i] = b[i];
}
extern "C" double hostcheck_fprati(double* pf /* p1 f1 p2 f2 p3 f3, p1 f1 p3 f3 updated */) {
  return fitpack::fprati(pf[0], pf[1], pf[2], pf[3], pf[4], pf[5]);
}
extern "C" void hostcheck_fpknot(int nest, const double* x, int* n, double* t, double* fpint, int* nrdata, int* nrint) {
  std::vector<double> tv(t, t + nest), fv(fpint, fpint + nest);
  std::vector<int> nd(nrdata, nrdata + nest);
  fitpack::fpknot(x, tv, *n, fv, nd, *nrint);
  for (int i = 0; i < nest; ++i) { t[i] = tv[i]; fpint[i] = fv[i]; nrdata[i] = nd[i]; }
}
extern "C" void hostcheck_fpknot_batch(int nest, const double* x, int* n, double* t, double* fpint, int* nrdata, int* nrint, int nplus, int nmax) {
  std::vector<double> tv(t, t + nest), fv(fpint, fpint + nest);
  std::vector<int> nd(nrdata, nrdata + nest);
  fitpack::fpknot_batch(x, tv, *n, fv, nd, *nrint, nplus, nmax, nest);
  for (int i = 0; i < nest; ++i) { t[i] = tv[i]; fpint[i] = fv[i]; nrdata[i] = nd[i]; }
}
// op: 0 a+b, 1 a*b, 2 a/b, 3 sqrt(a), 4 exact product of the two high words;  a, b, out: (hi, lo)
extern "C" void hostcheck_dd(int op, const double* a, const double* b, double* out) {
  const dd x(a[0], a[1]), y(b[0], b[1]);
  dd r;
  switch (op) {
    case 0: r = x + y; break;
    case 1: r = x * y; break;
    case 2: r = x / y; break;
    case 3: r = num_sqrt(x); break;
    default: r = dd_two_prod(a[0], b[0]); break;
  }
  out[0] = r.hi; out[1] = r.lo;
}

// host build of the PnP math (mvus_amd/csrc/pnp.hip.h)
extern "C" int hostcheck_pnp_dlt6(const double* Xs /* [6][3] */, const double* xn /* [6][2] */, double* R, double* t) {
  return pnp_dlt6(reinterpret_cast<const double (*)[3]>(Xs), reinterpret_cast<const double (*)[2]>(xn), R, t) ? 1 : 0;
}
extern "C" int hostcheck_pnp_project(const double* K, const double* d, const double* R, const double* t, const double* X, double* uv) {
  return pnp_project(K, d, R, t, X, uv[0], uv[1]) ? 1 : 0;
}
extern "C" int hostcheck_pnp_point_normal(const double* K, const double* d, const double* R, const double* t, const double* X, double u, double v, double* acc28) {
  for (int k = 0; k < 28; ++k) acc28[k] = 0.0;
  return pnp_point_normal(K, d, R, t, X, u, v, acc28) ? 1 : 0;
}
extern "C" void hostcheck_pnp_sample6(unsigned long long seed, int h, long long N, long long* idx) { pnp_sample6(seed, h, N, idx); }
extern "C" void hostcheck_rotation_to_rvec(const double* R, double* r) { rotation_to_rvec(R, r); }
