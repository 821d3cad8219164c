// TEST-ONLY plain-C++ backend for the templated optimiser (mvus_amd/csrc/ba_solver.h).
//
// It evaluates residuals / Jacobians with the same ba_math.h functions the HIP kernels call and
// stores J in the same slot layout, so the CPU test-suite can check the restated scipy optimiser,
// the block-sparse J v / J^T u operators, the pattern mask and the normal-equation algebra against
// the oracle without a GPU.  It is compiled into tests/hostcheck/libhostcheck.so only -- the
// product library libmvusba.so has no CPU path.
#include "../../mvus_amd/csrc/ba_partition.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../mvus_amd/csrc/ba_problem.h"
#include "../../mvus_amd/csrc/ba_solver.h"
#ifdef MVUS_WITH_SCHUR
#include "../../mvus_amd/csrc/ba_schur.h"
#endif

using namespace mvus;

namespace {

struct HostBackend {
  HostProblem hp;
  std::vector<double> J, mJ, u_obs, v_obs;
  std::vector<int32_t> span, pat0, mctrl;
  bool has_pattern = false, pattern_uploaded = false;
  std::vector<int32_t> ms_pat_canon;
  std::vector<int32_t> fd_groups;
  int fd_ngroups = 0;

  int64_t n() const { return hp.n; }
  int64_t m_local() const { return hp.m; }
  int64_t m_global() const { return hp.m; }
  double* alloc(int64_t len) { return new double[len > 0 ? len : 1](); }
  void release(double* p) { delete[] p; }
  void upload(double* d, const double* s, int64_t len) { std::memcpy(d, s, sizeof(double) * len); }
  void download(double* d, const double* s, int64_t len) { std::memcpy(d, s, sizeof(double) * len); }
  void copy(double* d, const double* s, int64_t len) { std::memcpy(d, s, sizeof(double) * len); }
  void fill(double* d, double v, int64_t len) { for (int64_t i = 0; i < len; ++i) d[i] = v; }
  void axpby(int64_t len, double a, const double* x, double b, const double* y, double* out) {
    for (int64_t i = 0; i < len; ++i) out[i] = a * x[i] + b * y[i];
  }
  void mul(int64_t len, const double* x, const double* y, double* out) { for (int64_t i = 0; i < len; ++i) out[i] = x[i] * y[i]; }
  // MVUS_HOST_EXACT_SUMS=1 (an experiment of tools/micro/rsf_clusters.py): every long sum of the solver -- dot
  // products, J v, J^T u -- accumulated in 80-bit long double (x87: 64-bit mantissa, products exact) and rounded once
  const bool exact_sums = std::getenv("MVUS_HOST_EXACT_SUMS") != nullptr;
  double dot_n(const double* a, const double* b, int64_t len) {
    if (exact_sums) { long double s = 0; for (int64_t i = 0; i < len; ++i) s += (long double)a[i] * (long double)b[i]; return (double)s; }
    double s = 0; for (int64_t i = 0; i < len; ++i) s += a[i] * b[i]; return s;
  }
  double dot_m(const double* a, const double* b) { return dot_n(a, b, hp.m); }
  // device-resident LM driver (ba_schur.h): host memory plays the role of device memory
  std::vector<double> lbv, ubv;
  double lm_s[8];
  void set_bounds(const std::vector<double>& lb, const std::vector<double>& ub) { lbv = lb; ubv = ub; }
  const double* lb_ptr() const { return lbv.data(); }
  const double* ub_ptr() const { return ubv.data(); }
  double* lm_scalars() { return lm_s; }
  void dot_m_into(const double* a, const double* b, double* out) { *out = dot_m(a, b); }
  bool residual_sq(const double* x, double* f, double* out, double* = nullptr, int64_t = 0) { residual(x, f); dot_m_into(f, f, out); return false; }
  void lm_gnorm(const double* x, const double* lb, const double* ub, const double* g, double* out) { *out = mvus::lm_gnorm_host(hp.n, x, lb, ub, g); }
  void lm_trial(const double* x, const double* p, const double* lb, const double* ub, const double* g, const double* D,
                const int* fail, double* x_new, double* out, double* gn, double*, double* pn2 = nullptr, double delta = 0.0, const double* = nullptr) {
    double cut = 1.0;
    if (pn2) { double s = 0; for (int64_t i = 0; i < hp.n; ++i) s += p[i] * p[i]; *pn2 = s; if (delta > 0 && s > delta * delta) cut = delta / std::sqrt(s); }
    mvus::lm_trial_host(hp.n, x, p, lb, ub, g, D, *fail, x_new, out, gn, cut);
  }
  static constexpr bool kSwapResiduals = false;
  static constexpr bool kDeviceLsmr = false;
  std::vector<double> lm_x[2];
  double* lm_xbuf(int k) { lm_x[k].resize((size_t)hp.n); return lm_x[k].data(); }
  int lm_resume(const double*) { return -1; }
  bool reshard_pending() { return false; }
  void lm_remember(const double*, const double*, const double*) {}
  mvus::LmCarry lm_carry(int) { return {}; }
  void lm_keep(const mvus::LmCarry&, int) {}
  void fetch_mark() {}
  void fetch_enqueue(const double*, int) {}
  double* mirror_dev(int) { return nullptr; }
  const double* mirror_host(int) const { return nullptr; }
  void adopt_residual(double*&, double*&) {}
  void fetch(const double* src, int k, double* host) { std::memcpy(host, src, sizeof(double) * k); }

  void init() {
    J.assign((size_t)2 * hp.NS * hp.M, 0.0); span.assign(hp.M, -1); pat0.assign(hp.M, -1);
    mJ.assign((size_t)36 * hp.T, 0.0); mctrl.assign((size_t)3 * hp.T, -1);
    u_obs = hp.u_raw; v_obs = hp.v_raw;
    if (!hp.calib && hp.undist)
      for (int c = 0; c < hp.C; ++c) {
        const double fx = hp.K[4 * c], fy = hp.K[4 * c + 1], cx = hp.K[4 * c + 2], cy = hp.K[4 * c + 3];
        for (int64_t i = hp.det_off[c]; i < hp.det_off[c + 1]; ++i) {
          double xn, yn;
          undistort5<false>((hp.u_raw[i] - cx) / fx, (hp.v_raw[i] - cy) / fy, &hp.dist[5 * c], xn, yn, nullptr, nullptr);
          u_obs[i] = fx * xn + cx; v_obs[i] = fy * yn + cy;
        }
      }
  }

  template <bool JAC>
  void eval(const double* x, double* f, int jac_mode) {
    const SplineView sp = hp.spline_view();
    const int NS = hp.NS;
    std::vector<double> jx(NS), jy(NS);
    for (int c = 0; c < hp.C; ++c) {
      CamState cam;
      load_cam_state(x, hp.C, c, hp.calib, hp.K.data(), hp.dist.data(), hp.H[c], cam);
      const int64_t a = hp.det_off[c], Mc = hp.det_off[c + 1] - a;
      for (int64_t i = a; i < a + Mc; ++i) {
        ObsResult r = hp.calib ? eval_observation<true, JAC>(cam, sp, x, hp.undist, hp.rs_free, hp.sync_free, hp.frame[i], hp.u_raw[i], hp.v_raw[i], u_obs[i], v_obs[i], jx.data(), jy.data())
                               : eval_observation<false, JAC>(cam, sp, x, hp.undist, hp.rs_free, hp.sync_free, hp.frame[i], hp.u_raw[i], hp.v_raw[i], u_obs[i], v_obs[i], jx.data(), jy.data());
        f[2 * a + (i - a)] = r.ex;
        f[2 * a + Mc + (i - a)] = r.ey;
        if (JAC) {
          int32_t ctrl = r.ctrl;
          if (jac_mode == MVUS_JAC_PATTERN && ctrl >= 0) {
            if (pat0[i] < 0) ctrl = -1;
            else mask_to_pattern(jx.data(), jy.data(), 3 + hp.P, ctrl, pat0[i]);
          }
          span[i] = ctrl;
          for (int k = 0; k < NS; ++k) {
            J[(size_t)k * hp.M + i] = ctrl >= 0 ? jx[k] : 0.0;
            J[(size_t)(NS + k) * hp.M + i] = ctrl >= 0 ? jy[k] : 0.0;
          }
        }
      }
    }
    const MotionView mv = hp.motion_view();
    for (int j = 0; j < hp.T; ++j) {
      double jrow[36]; int32_t cidx[3];
      f[2 * hp.M + j] = eval_motion_row<JAC>(mv, x, j, jac_mode == MVUS_JAC_PATTERN, jrow, cidx);
      if (JAC) {
        for (int k = 0; k < 36; ++k) mJ[(size_t)k * hp.T + j] = jrow[k];
        for (int k = 0; k < 3; ++k) mctrl[(size_t)k * hp.T + j] = cidx[k];
      }
    }
  }
  void residual(const double* x, double* f) { eval<false>(x, f, 0); }
  void jacobian(const double* x, double* f, int jac_mode) {
    if (jac_mode == MVUS_JAC_FD) jacobian_fd(x, f); else eval<true>(x, f, jac_mode);
  }
  void jacobian_fd(const double* x, double* f) {
    const int64_t n = hp.n, m = hp.m;
    const int NS = hp.NS, B = 3 + hp.P;
    eval<false>(x, f, 0);
    std::vector<double> h(n), dx(n), xg(n), F((size_t)fd_ngroups * m);
    for (int64_t j = 0; j < n; ++j) {
      const bool bounded = hp.rs_bounds && j >= 2 * hp.C && j < 3 * hp.C;
      h[j] = fd_step(x[j], bounded ? 0.0 : -INFINITY, bounded ? 1.0 : INFINITY);
      dx[j] = (x[j] + h[j]) - x[j];
    }
    for (int g = 0; g < fd_ngroups; ++g) {
      for (int64_t j = 0; j < n; ++j) xg[j] = fd_groups[j] == g ? x[j] + h[j] : x[j];
      eval<false>(xg.data(), &F[(size_t)g * m], 0);
    }
    for (int c = 0; c < hp.C; ++c) {
      const int64_t a = hp.det_off[c], Mc = hp.det_off[c + 1] - a;
      for (int64_t i = a; i < a + Mc; ++i) {
        const int p = pat0[i];
        const int base = p >= 0 ? pattern_fd_base(p, hp.N, hp.ctrl_x0.data()) : p;
        span[i] = base;
        const int64_t rx = 2 * a + (i - a), ry = rx + Mc;
        for (int k = 0; k < NS; ++k) {
          int col = -1;
          if (p >= 0) {
            if (k < B) { if (!(k == 2 && !hp.rs_free) && !(k < 2 && !hp.sync_free)) col = col_of(c, k); }
            else { const int q = (k - B) / 3, d = (k - B) % 3; if (pattern_has(p, base + q)) col = hp.ctrl_x0[base] + q + d * hp.ctrl_stride[base]; }
          }
          J[(size_t)k * hp.M + i] = col >= 0 ? (F[(size_t)fd_groups[col] * m + rx] - f[rx]) / dx[col] : 0.0;
          J[(size_t)(NS + k) * hp.M + i] = col >= 0 ? (F[(size_t)fd_groups[col] * m + ry] - f[ry]) / dx[col] : 0.0;
        }
      }
    }
    for (int j = 0; j < hp.T; ++j) {
      for (int k = 0; k < 36; ++k) mJ[(size_t)k * hp.T + j] = 0.0;
      mctrl[j] = -1; mctrl[(size_t)2 * hp.T + j] = -1; mctrl[(size_t)hp.T + j] = -1;
      if (hp.ms_part[j] < 0) continue;
      const int pc = hp.ms_pat[j];
      const int base = pattern_fd_base(pc, hp.N, hp.ctrl_x0.data());
      mctrl[(size_t)hp.T + j] = base;
      const int64_t row = 2 * hp.M + j;
      for (int q = 0; q < 4; ++q) {
        if (!pattern_has(pc, base + q)) continue;
        for (int d = 0; d < 3; ++d) {
          const int col = hp.ctrl_x0[base] + q + d * hp.ctrl_stride[base];
          mJ[(size_t)(12 + 3 * q + d) * hp.T + j] = (F[(size_t)fd_groups[col] * m + row] - f[row]) / dx[col];
        }
      }
    }
  }

  void set_pattern(const double* x0) {
    const SplineView sp = hp.spline_view();
    for (int c = 0; c < hp.C; ++c) {
      CamState cam;
      load_cam_state(x0, hp.C, c, hp.calib, hp.K.data(), hp.dist.data(), hp.H[c], cam);
      for (int64_t i = hp.det_off[c]; i < hp.det_off[c + 1]; ++i)
        pat0[i] = observation_pattern(cam, sp, hp.frame[i], hp.v_raw[i]);
    }
    if (pattern_uploaded) hp.ms_pat = ms_pat_canon;
    has_pattern = true; pattern_uploaded = false;
  }
  void upload_pattern(const int32_t* pat, const int32_t* mpat) {
    pat0.assign(pat, pat + hp.M);
    if (!pattern_uploaded) ms_pat_canon = hp.ms_pat;
    if (mpat && hp.T > 0) hp.ms_pat.assign(mpat, mpat + hp.T);
    has_pattern = true; pattern_uploaded = true;
  }

  int col_of(int c, int k) const { return k < 3 ? k * hp.C + c : 3 * hp.C + c * hp.P + (k - 3); }

  template <class T>
  void jv_t(const double* v, double* y) {
    const int NS = hp.NS, B = 3 + hp.P;
    for (int c = 0; c < hp.C; ++c) {
      const int64_t a = hp.det_off[c], Mc = hp.det_off[c + 1] - a;
      for (int64_t i = a; i < a + Mc; ++i) {
        T sx = 0, sy = 0;
        const int g = span[i];
        if (g >= 0) {
          for (int k = 0; k < B; ++k) { const T vv = v[col_of(c, k)]; sx += (T)J[(size_t)k * hp.M + i] * vv; sy += (T)J[(size_t)(NS + k) * hp.M + i] * vv; }
          const int x0 = hp.ctrl_x0[g], st = hp.ctrl_stride[g];
          for (int q = 0; q < 4; ++q)
            for (int d = 0; d < 3; ++d) {
              const T vv = v[x0 + q + d * st];
              sx += (T)J[(size_t)(B + 3 * q + d) * hp.M + i] * vv; sy += (T)J[(size_t)(NS + B + 3 * q + d) * hp.M + i] * vv;
            }
        }
        y[2 * a + (i - a)] = (double)sx; y[2 * a + Mc + (i - a)] = (double)sy;
      }
    }
    for (int j = 0; j < hp.T; ++j) {
      T s = 0;
      for (int k = 0; k < 3; ++k) {
        const int g = mctrl[(size_t)k * hp.T + j];
        if (g < 0) continue;
        const int x0 = hp.ctrl_x0[g], st = hp.ctrl_stride[g];
        for (int q = 0; q < 4; ++q) for (int d = 0; d < 3; ++d) s += (T)mJ[(size_t)(12 * k + 3 * q + d) * hp.T + j] * (T)v[x0 + q + d * st];
      }
      y[2 * hp.M + j] = (double)s;
    }
  }
  template <class T>
  void jtu_t(const double* u, double* zout) {
    const int NS = hp.NS, B = 3 + hp.P;
    std::vector<T> z((size_t)hp.n, (T)0);
    for (int c = 0; c < hp.C; ++c) {
      const int64_t a = hp.det_off[c], Mc = hp.det_off[c + 1] - a;
      for (int64_t i = a; i < a + Mc; ++i) {
        const int g = span[i];
        if (g < 0) continue;
        const T ux = u[2 * a + (i - a)], uy = u[2 * a + Mc + (i - a)];
        for (int k = 0; k < B; ++k) z[col_of(c, k)] += (T)J[(size_t)k * hp.M + i] * ux + (T)J[(size_t)(NS + k) * hp.M + i] * uy;
        const int x0 = hp.ctrl_x0[g], st = hp.ctrl_stride[g];
        for (int q = 0; q < 4; ++q)
          for (int d = 0; d < 3; ++d)
            z[x0 + q + d * st] += (T)J[(size_t)(B + 3 * q + d) * hp.M + i] * ux + (T)J[(size_t)(NS + B + 3 * q + d) * hp.M + i] * uy;
      }
    }
    for (int j = 0; j < hp.T; ++j)
      for (int k = 0; k < 3; ++k) {
        const int g = mctrl[(size_t)k * hp.T + j];
        if (g < 0) continue;
        const int x0 = hp.ctrl_x0[g], st = hp.ctrl_stride[g];
        for (int q = 0; q < 4; ++q) for (int d = 0; d < 3; ++d) z[x0 + q + d * st] += (T)mJ[(size_t)(12 * k + 3 * q + d) * hp.T + j] * (T)u[2 * hp.M + j];
      }
    for (int64_t i = 0; i < hp.n; ++i) zout[i] = (double)z[i];
  }

  void jv(const double* v, double* y) {
    if (exact_sums) { jv_t<long double>(v, y); return; }
    const int NS = hp.NS, B = 3 + hp.P;
    for (int c = 0; c < hp.C; ++c) {
      const int64_t a = hp.det_off[c], Mc = hp.det_off[c + 1] - a;
      for (int64_t i = a; i < a + Mc; ++i) {
        double sx = 0, sy = 0;
        const int g = span[i];
        if (g >= 0) {
          for (int k = 0; k < B; ++k) { const double vv = v[col_of(c, k)]; sx += J[(size_t)k * hp.M + i] * vv; sy += J[(size_t)(NS + k) * hp.M + i] * vv; }
          const int x0 = hp.ctrl_x0[g], st = hp.ctrl_stride[g];
          for (int q = 0; q < 4; ++q)
            for (int d = 0; d < 3; ++d) {
              const double vv = v[x0 + q + d * st];
              sx += J[(size_t)(B + 3 * q + d) * hp.M + i] * vv; sy += J[(size_t)(NS + B + 3 * q + d) * hp.M + i] * vv;
            }
        }
        y[2 * a + (i - a)] = sx; y[2 * a + Mc + (i - a)] = sy;
      }
    }
    for (int j = 0; j < hp.T; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) {
        const int g = mctrl[(size_t)k * hp.T + j];
        if (g < 0) continue;
        const int x0 = hp.ctrl_x0[g], st = hp.ctrl_stride[g];
        for (int q = 0; q < 4; ++q) for (int d = 0; d < 3; ++d) s += mJ[(size_t)(12 * k + 3 * q + d) * hp.T + j] * v[x0 + q + d * st];
      }
      y[2 * hp.M + j] = s;
    }
  }

  void jtu(const double* u, double* z) {
    if (exact_sums) { jtu_t<long double>(u, z); return; }
    const int NS = hp.NS, B = 3 + hp.P;
    for (int64_t i = 0; i < hp.n; ++i) z[i] = 0;
    for (int c = 0; c < hp.C; ++c) {
      const int64_t a = hp.det_off[c], Mc = hp.det_off[c + 1] - a;
      for (int64_t i = a; i < a + Mc; ++i) {
        const int g = span[i];
        if (g < 0) continue;
        const double ux = u[2 * a + (i - a)], uy = u[2 * a + Mc + (i - a)];
        for (int k = 0; k < B; ++k) z[col_of(c, k)] += J[(size_t)k * hp.M + i] * ux + J[(size_t)(NS + k) * hp.M + i] * uy;
        const int x0 = hp.ctrl_x0[g], st = hp.ctrl_stride[g];
        for (int q = 0; q < 4; ++q)
          for (int d = 0; d < 3; ++d)
            z[x0 + q + d * st] += J[(size_t)(B + 3 * q + d) * hp.M + i] * ux + J[(size_t)(NS + B + 3 * q + d) * hp.M + i] * uy;
      }
    }
    for (int j = 0; j < hp.T; ++j)
      for (int k = 0; k < 3; ++k) {
        const int g = mctrl[(size_t)k * hp.T + j];
        if (g < 0) continue;
        const int x0 = hp.ctrl_x0[g], st = hp.ctrl_stride[g];
        for (int q = 0; q < 4; ++q) for (int d = 0; d < 3; ++d) z[x0 + q + d * st] += mJ[(size_t)(12 * k + 3 * q + d) * hp.T + j] * u[2 * hp.M + j];
      }
  }
};

thread_local std::string g_err;

}  // namespace

extern "C" {

const char* hostcheck_error() { return g_err.c_str(); }

void* hostcheck_create(const mvus_problem* p) {
  HostBackend* be = new HostBackend();
  g_err = be->hp.build(p);
  if (!g_err.empty()) { delete be; return nullptr; }
  be->init();
  return be;
}
void hostcheck_destroy(void* h) { delete static_cast<HostBackend*>(h); }
int64_t hostcheck_n(void* h) { return static_cast<HostBackend*>(h)->hp.n; }
int64_t hostcheck_m(void* h) { return static_cast<HostBackend*>(h)->hp.m; }
int hostcheck_T(void* h) { return static_cast<HostBackend*>(h)->hp.T; }

int hostcheck_residual(void* h, const double* x, double* f) { static_cast<HostBackend*>(h)->residual(x, f); return 0; }

int hostcheck_set_pattern(void* h, const double* x0, int32_t* pat) {
  HostBackend* be = static_cast<HostBackend*>(h);
  be->set_pattern(x0);
  if (pat) std::memcpy(pat, be->pat0.data(), sizeof(int32_t) * be->hp.M);
  return 0;
}

// dense Jacobian (m x n, row-major) of the current mode, for comparison with scipy / finite differences
int hostcheck_dense_jacobian(void* h, const double* x, int jac_mode, double* f, double* Jd) {
  HostBackend* be = static_cast<HostBackend*>(h);
  be->jacobian(x, f, jac_mode);
  const int64_t n = be->hp.n, m = be->hp.m;
  std::vector<double> e(n, 0.0), col(m);
  for (int64_t j = 0; j < n; ++j) {
    e[j] = 1.0;
    be->jv(e.data(), col.data());
    e[j] = 0.0;
    for (int64_t i = 0; i < m; ++i) Jd[i * n + j] = col[i];
  }
  return 0;
}

int hostcheck_upload_pattern(void* h, const int32_t* pat, const int32_t* mpat) { static_cast<HostBackend*>(h)->upload_pattern(pat, mpat); return 0; }
int hostcheck_motion_pattern(void* h, int32_t* mpat) {
  HostBackend* be = static_cast<HostBackend*>(h);
  if (be->hp.T > 0) std::memcpy(mpat, be->hp.ms_pat.data(), sizeof(int32_t) * be->hp.T);
  return 0;
}

int hostcheck_set_fd_groups(void* h, const int32_t* groups, int ngroups) {
  HostBackend* be = static_cast<HostBackend*>(h);
  be->fd_groups.assign(groups, groups + be->hp.n);
  be->fd_ngroups = ngroups;
  return 0;
}

int hostcheck_jtu(void* h, const double* u, double* z) { static_cast<HostBackend*>(h)->jtu(u, z); return 0; }

int hostcheck_solve(void* h, double* x, const mvus_solve_opts* o, mvus_result* r, double* f_out) {
  HostBackend* be = static_cast<HostBackend*>(h);
  const int64_t n = be->hp.n;
  std::vector<double> xv(x, x + n), lb(n, -INFINITY), ub(n, INFINITY);
  if (be->hp.rs_bounds) for (int c = 0; c < be->hp.C; ++c) { lb[2 * be->hp.C + c] = 0.0; ub[2 * be->hp.C + c] = 1.0; }
  SolveOptions so;
  so.jac_mode = o->jac_mode; so.max_nfev = o->max_nfev; so.ftol = o->ftol; so.xtol = o->xtol; so.gtol = o->gtol;
  so.lsmr_atol = o->lsmr_atol; so.lsmr_btol = o->lsmr_btol; so.lsmr_conlim = o->lsmr_conlim; so.lsmr_maxiter = o->lsmr_maxiter; so.verbose = o->verbose; so.lm_lambda_min = o->lm_lambda_min; so.lm_trust_radius = o->lm_trust_radius;
  if (so.jac_mode == MVUS_JAC_PATTERN && !be->pattern_uploaded) be->set_pattern(x);
  std::vector<double> f(be->hp.m);
  SolveResult sr;
#ifdef MVUS_WITH_SCHUR
  if (o->solver == MVUS_SOLVER_LM_SCHUR) { HostSchur sc; sr = lm_schur(*be, sc, xv.data(), lb, ub, so, f.data()); }
  else
#endif
    sr = trf_lsmr(*be, xv, lb, ub, so, f.data());
  if (sr.error) { g_err = "non-finite residuals at x0 or x0 outside bounds"; return sr.error; }
  std::memcpy(x, xv.data(), sizeof(double) * n);
  if (f_out) std::memcpy(f_out, f.data(), sizeof(double) * be->hp.m);
  r->cost = sr.cost; r->optimality = sr.optimality; r->nfev = sr.nfev; r->njev = sr.njev; r->status = sr.status;
  r->lin_iters = sr.lin_iters; r->initial_cost = sr.initial_cost; r->solve_ms = 0;
  return 0;
}

}  // extern "C"

// LSMR on the Jacobian currently held (after hostcheck_dense_jacobian): min |J x - b|^2 + damp^2 |x|^2
extern "C" int hostcheck_lsmr(void* h, const double* b, double damp, double atol, double btol, double conlim,
                              int maxiter, double* x_out, int* itn) {
  HostBackend* be = static_cast<HostBackend*>(h);
  Lsmr<HostBackend> l(*be);
  return l.run(nullptr, nullptr, b, damp, atol, btol, conlim, maxiter > 0 ? maxiter : std::min(be->hp.m, be->hp.n), x_out, itn);
}

// partition of a control-point chain (ba_partition.h): returns the number of interiors, fills i0/i1 (scalar rows) and the
// separators' first rows; *nsep receives their count
extern "C" int hostcheck_partition(int c0, int n, int sctrl, int close, int len, int* i0, int* i1, int* sep, int* nsep) {
  const mvus::ChainPart cp = mvus::partition_chain(c0, n, sctrl, close != 0, len);
  for (size_t k = 0; k < cp.i0.size(); ++k) { i0[k] = cp.i0[k]; i1[k] = cp.i1[k]; }
  for (size_t k = 0; k < cp.sep.size(); ++k) sep[k] = cp.sep[k];
  *nsep = (int)cp.sep.size();
  return (int)cp.i0.size();
}

extern "C" void hostcheck_tr2d(const double* B, const double* g, double Delta, double* p) {
  mvus::detail::solve_trust_region_2d(B, g, Delta, p);
}
