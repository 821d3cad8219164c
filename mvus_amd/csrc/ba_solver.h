// Trust-region-reflective + LSMR optimiser over an abstract backend.
//
// Restates, for the call the reference makes at multiviewunsynch/reconstruction/common.py:670
//     least_squares(fn, x0, jac_sparsity=A, tr_solver='lsmr', xtol=1e-12, max_nfev=max_iter, bounds=...)
// the algorithm of scipy 1.15.3 (third-party, not vendored by the reference):
//     scipy/optimize/_lsq/trf.py      trf_no_bounds (:401-560), trf_bounds (:205-395), select_step (:128-202)
//     scipy/optimize/_lsq/common.py   solve_trust_region_2d (:171), update_tr_radius (:222),
//                                     build_quadratic_1d (:251), minimize_quadratic_1d (:302),
//                                     step_size_to_bound (:372), make_strictly_feasible (:440),
//                                     CL_scaling_vector (:467), check_termination (:705)
//     scipy/sparse/linalg/_isolve/lsmr.py   lsmr (:29-480), _sym_ortho
// with one deliberate difference: the Jacobian is analytic (optionally masked to the reference's
// sparsity pattern) instead of sparse 2-point finite differences.
//
// All O(m) and O(nnz(J)) work goes through the Backend (HIP kernels in the product, a plain-C++
// backend in tests/hostcheck); the O(n) trust-region logic runs on the host.
//
// Backend concept (B):
//   int64_t n(), m_local(), m_global();  double* alloc(len); void release(p);
//   void upload(dst_dev, src_host, len); void download(dst_host, src_dev, len); void copy(dst, src, len);
//   void fill(dst, value, len); void axpby(len, a, x, b, y, out); void mul(len, x, y, out);
//   double dot_n(a, b, len);   // replicated vectors
//   double dot_m(a, b);        // row-sharded vectors of length m_local(), summed over ranks
//   void residual(x_dev, f_dev); void jacobian(x_dev, f_dev, jac_mode);
//   void jv(v_dev, y_dev);     // y[m_local] = J v
//   void jtu(u_dev, z_dev);    // z[n] = J^T u summed over ranks
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

namespace mvus {

struct SolveOptions {
  int jac_mode = 0;
  int max_nfev = 10;
  double ftol = 1e-8, xtol = 1e-12, gtol = 1e-8;
  double lsmr_atol = 1e-6, lsmr_btol = 1e-6, lsmr_conlim = 1e8;
  int lsmr_maxiter = 0;
  int verbose = 0;
  double lm_lambda_min = 3e-3;   // floor of the LM damping (see mvus_solve_opts)
  double lm_trust_radius = -1;   // < 0: none; 0: Delta_0 = |x0| (scipy); > 0: Delta_0 (see mvus_solve_opts)
  double lm_lambda0 = 0;   // > 0: initial LM damping (a handle carries it over from its previous solve)
  double lm_nu0 = 0;       // > 0: initial damping growth factor (carried with it, so that a run of one-trial solves
                           // escalates the damping like one long solve does)
};

struct SolveResult {
  double cost = 0, optimality = 0, initial_cost = 0;
  int nfev = 0, njev = 0, status = 0, lin_iters = 0;
  int error = 0;  // 0 ok, -3 numeric (non-finite f0 / infeasible x0)
  double lm_lambda = 0;  // final LM damping
  double lm_nu = 2;      // final growth factor of the damping
  bool jac_stale = false;  // the Jacobian held by the backend belongs to an earlier point than the returned x
  bool async_tail = false; // LM_SCHUR: a speculative linearisation at the returned point may still be running on the stream (its results
                           // are only ever read by later work on the same stream: the caller need not wait for it)
};

// What an LM solve leaves behind for the point it returned, for a caller that continues from exactly that point (ba_schur.h):
// f(x) in the backend's residual buffer with its cost, and -- when the solve's last act was a linearisation there -- the normal equations.
struct LmCarry {
  bool f_valid = false, lin_valid = false;
  double cost = 0;
};

// returns the pooled work vectors of a solve when the driver leaves, by return or by exception
template <class B>
struct PoolGuard {
  B& be;
  std::vector<double*> bufs;
  explicit PoolGuard(B& b) : be(b) {}
  double* get(int64_t len) { double* p = be.alloc(len); bufs.push_back(p); return p; }
  void replace(double* held, double* now) { for (double*& p : bufs) if (p == held) { p = now; return; } }   // a held buffer changed roles with one outside
  ~PoolGuard() { for (double* p : bufs) be.release(p); }
  PoolGuard(const PoolGuard&) = delete;
  PoolGuard& operator=(const PoolGuard&) = delete;
};

namespace detail {

inline double sgn(double a) { return (a > 0) - (a < 0); }

// scipy.sparse.linalg._isolve.lsmr._sym_ortho
inline void sym_ortho(double a, double b, double& c, double& s, double& r) {
  if (b == 0) { c = sgn(a); s = 0; r = std::fabs(a); }
  else if (a == 0) { c = 0; s = sgn(b); r = std::fabs(b); }
  else if (std::fabs(b) > std::fabs(a)) {
    const double tau = a / b;
    s = sgn(b) / std::sqrt(1 + tau * tau);
    c = s * tau;
    r = b / s;
  } else {
    const double tau = b / a;
    c = sgn(a) / std::sqrt(1 + tau * tau);
    s = c * tau;
    r = a / c;
  }
}

inline double norm2(const std::vector<double>& a) { double s = 0; for (double v : a) s += v * v; return std::sqrt(s); }
inline double dot(const std::vector<double>& a, const std::vector<double>& b) { double s = 0; for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i]; return s; }

// real roots of c[0] t^d + ... + c[d] (numpy.roots semantics: leading zeros stripped)
inline std::vector<double> real_roots(std::vector<double> c) {
  std::vector<double> out;
  size_t lead = 0;
  while (lead < c.size() && c[lead] == 0.0) ++lead;
  c.erase(c.begin(), c.begin() + lead);
  while (!c.empty() && c.back() == 0.0) { out.push_back(0.0); c.pop_back(); }
  const int d = (int)c.size() - 1;
  if (d < 1) return out;
  if (d == 1) { out.push_back(-c[1] / c[0]); return out; }
  using cd = std::complex<double>;
  std::vector<double> a(c.size());
  for (size_t i = 0; i < c.size(); ++i) a[i] = c[i] / c[0];
  double rad = 0;
  for (int i = 1; i <= d; ++i) rad = std::max(rad, std::pow(std::fabs(a[i]), 1.0 / i));
  rad = 2.0 * rad + 1e-300;
  std::vector<cd> z(d);
  for (int k = 0; k < d; ++k) z[k] = std::polar(rad * (0.5 + 0.5 * (k + 1) / d), 2.0 * M_PI * k / d + 0.4);
  auto eval = [&](cd t, cd& p, cd& dp) { p = a[0]; dp = 0; for (int i = 1; i <= d; ++i) { dp = dp * t + p; p = p * t + a[i]; } };
  for (int it = 0; it < 200; ++it) {
    double change = 0;
    for (int k = 0; k < d; ++k) {
      cd p, dp; eval(z[k], p, dp);
      if (std::abs(p) == 0) continue;
      cd ratio = p / dp, sum = 0;
      for (int j = 0; j < d; ++j) if (j != k) sum += 1.0 / (z[k] - z[j]);
      cd w = ratio / (1.0 - ratio * sum);
      z[k] -= w;
      change = std::max(change, std::abs(w) / (1e-300 + std::abs(z[k])));
    }
    if (change < 1e-15) break;
  }
  for (int k = 0; k < d; ++k) {
    if (std::fabs(z[k].imag()) <= 1e-7 * (1.0 + std::fabs(z[k].real()))) {
      double t = z[k].real();
      for (int it = 0; it < 3; ++it) {  // Newton polish on the real polynomial
        double p = a[0], dp = 0;
        for (int i = 1; i <= d; ++i) { dp = dp * t + p; p = p * t + a[i]; }
        if (dp == 0) break;
        t -= p / dp;
      }
      out.push_back(t);
    }
  }
  return out;
}

// scipy solve_trust_region_2d (common.py:171-219)
inline void solve_trust_region_2d(const double B[4], const double g[2], double Delta, double p[2]) {
  // cho_factor on the 2x2
  if (B[0] > 0) {
    const double l00 = std::sqrt(B[0]), l10 = B[2] / l00, d11 = B[3] - l10 * l10;
    if (d11 > 0) {
      const double l11 = std::sqrt(d11);
      const double y0 = -g[0] / l00, y1 = (-g[1] - l10 * y0) / l11;
      const double p1 = y1 / l11, p0 = (y0 - l10 * p1) / l00;
      if (p0 * p0 + p1 * p1 <= Delta * Delta) { p[0] = p0; p[1] = p1; return; }
    }
  }
  const double a = B[0] * Delta * Delta, b = B[1] * Delta * Delta, c = B[3] * Delta * Delta;
  const double d = g[0] * Delta, f = g[1] * Delta;
  std::vector<double> t = real_roots({-b + d, 2 * (a - c + f), 6 * b, 2 * (-a + c + f), -b - d});
  double best = std::numeric_limits<double>::infinity();
  p[0] = 0; p[1] = -Delta;  // t -> inf limit, always a boundary point
  {
    const double q0 = 0, q1 = -Delta;
    best = 0.5 * (q0 * (B[0] * q0 + B[1] * q1) + q1 * (B[2] * q0 + B[3] * q1)) + g[0] * q0 + g[1] * q1;
  }
  for (double tt : t) {
    const double q0 = Delta * 2 * tt / (1 + tt * tt), q1 = Delta * (1 - tt * tt) / (1 + tt * tt);
    const double val = 0.5 * (q0 * (B[0] * q0 + B[1] * q1) + q1 * (B[2] * q0 + B[3] * q1)) + g[0] * q0 + g[1] * q1;
    if (val < best) { best = val; p[0] = q0; p[1] = q1; }
  }
}

inline void minimize_quadratic_1d(double a, double b, double lb, double ub, double c, double& t_out, double& y_out) {
  double ts[3] = {lb, ub, 0};
  int nt = 2;
  if (a != 0) {
    const double ext = -0.5 * b / a;
    if (lb < ext && ext < ub) ts[nt++] = ext;
  }
  t_out = ts[0]; y_out = ts[0] * (a * ts[0] + b) + c;
  for (int i = 1; i < nt; ++i) {
    const double y = ts[i] * (a * ts[i] + b) + c;
    if (y < y_out) { y_out = y; t_out = ts[i]; }
  }
}

inline void update_tr_radius(double& Delta, double actual, double predicted, double step_norm, bool bound_hit, double& ratio) {
  if (predicted > 0) ratio = actual / predicted;
  else if (predicted == 0 && actual == 0) ratio = 1;
  else ratio = 0;
  if (ratio < 0.25) Delta = 0.25 * step_norm;
  else if (ratio > 0.75 && bound_hit) Delta *= 2.0;
}

inline int check_termination(double dF, double F, double dx_norm, double x_norm, double ratio, double ftol, double xtol) {
  const bool f_ok = dF < ftol * F && ratio > 0.25;
  const bool x_ok = dx_norm < xtol * (xtol + x_norm);
  if (f_ok && x_ok) return 4;
  if (f_ok) return 2;
  if (x_ok) return 3;
  return -1;
}

inline void CL_scaling_vector(const std::vector<double>& x, const std::vector<double>& g, const std::vector<double>& lb,
                              const std::vector<double>& ub, std::vector<double>& v, std::vector<double>& dv) {
  const size_t n = x.size();
  v.assign(n, 1.0); dv.assign(n, 0.0);
  for (size_t i = 0; i < n; ++i) {
    if (g[i] < 0 && std::isfinite(ub[i])) { v[i] = ub[i] - x[i]; dv[i] = -1; }
    if (g[i] > 0 && std::isfinite(lb[i])) { v[i] = x[i] - lb[i]; dv[i] = 1; }
  }
}

inline bool in_bounds(const std::vector<double>& x, const std::vector<double>& lb, const std::vector<double>& ub) {
  for (size_t i = 0; i < x.size(); ++i) if (!(x[i] >= lb[i] && x[i] <= ub[i])) return false;
  return true;
}

inline double step_size_to_bound(const std::vector<double>& x, const std::vector<double>& s, const std::vector<double>& lb,
                                 const std::vector<double>& ub, std::vector<int>* hits) {
  const size_t n = x.size();
  std::vector<double> steps(n, std::numeric_limits<double>::infinity());
  double mn = std::numeric_limits<double>::infinity();
  for (size_t i = 0; i < n; ++i) {
    if (s[i] != 0) steps[i] = std::max((lb[i] - x[i]) / s[i], (ub[i] - x[i]) / s[i]);
    mn = std::min(mn, steps[i]);
  }
  if (hits) {
    hits->assign(n, 0);
    for (size_t i = 0; i < n; ++i) (*hits)[i] = (steps[i] == mn) ? (int)sgn(s[i]) : 0;
  }
  return mn;
}

inline void make_strictly_feasible(std::vector<double>& x, const std::vector<double>& lb, const std::vector<double>& ub, double rstep) {
  for (size_t i = 0; i < x.size(); ++i) {
    int active = 0;
    if (rstep == 0) {
      if (x[i] <= lb[i]) active = -1;
      if (x[i] >= ub[i]) active = 1;
    } else {
      const double ld = x[i] - lb[i], ud = ub[i] - x[i];
      const double lt = rstep * std::max(1.0, std::fabs(lb[i])), ut = rstep * std::max(1.0, std::fabs(ub[i]));
      if (std::isfinite(lb[i]) && ld <= std::min(ud, lt)) active = -1;
      if (std::isfinite(ub[i]) && ud <= std::min(ld, ut)) active = 1;
    }
    if (active == -1) x[i] = (rstep == 0) ? std::nextafter(lb[i], ub[i]) : lb[i] + rstep * std::max(1.0, std::fabs(lb[i]));
    if (active == 1) x[i] = (rstep == 0) ? std::nextafter(ub[i], lb[i]) : ub[i] - rstep * std::max(1.0, std::fabs(ub[i]));
    if (x[i] < lb[i] || x[i] > ub[i]) x[i] = 0.5 * (lb[i] + ub[i]);
  }
}

inline void intersect_trust_region(const std::vector<double>& x, const std::vector<double>& s, double Delta, double& t_neg, double& t_pos) {
  const double a = dot(s, s), b = dot(x, s), c = dot(x, x) - Delta * Delta;
  const double d = std::sqrt(std::max(b * b - a * c, 0.0));
  const double q = -(b + std::copysign(d, b));
  const double t1 = q / a, t2 = c / q;
  t_neg = std::min(t1, t2); t_pos = std::max(t1, t2);
}

}  // namespace detail

// ------------------------------------------------------------------------------------------------
// Scalar state of one LSMR run (scipy lsmr.py:316-470) and its two per-iteration updates, written once for the host loop
// and for the HIP backend's device-resident loop (k_lsmr_* in ba_kernels.hip.h), which runs whole batches of iterations
// without a host round trip.  No FMA contraction in here: host and device then produce the same bits.
// ------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
#define MVUS_SOLVER_HD __host__ __device__ inline
#else
#define MVUS_SOLVER_HD inline
#endif
struct LsmrScalars {
  double alpha = 0, beta = 0, zetabar = 0, alphabar = 0, rho = 1, rhobar = 1, cbar = 1, sbar = 0;
  double betadd = 0, betad = 0, rhodold = 1, tautildeold = 0, thetatilde = 0, zeta = 0, d = 0;
  double normA2 = 0, maxrbar = 0, minrbar = 1e+100, normb = 0, damp = 0, atol = 0, btol = 0, ctol = 0;
  double normr = 0, normar = 0, normA = 0, condA = 1;
  double c_hbar = 0, c_x = 0, c_h = 0;       // hbar = h + c_hbar hbar ; x += c_x hbar ; h = v + c_h h
  long long itn = 0, maxiter = 0;
  int istop = 0, pad = 0;
  double one = 1.0, zero = 0.0;               // read at run time by the device kernels: `a * x + b * y` with a = 1 must round like k_axpby's
};
namespace detail {
MVUS_SOLVER_HD double hd_sgn(double a) { return (a > 0) - (a < 0); }
MVUS_SOLVER_HD void hd_sym_ortho(double a, double b, double& c, double& s, double& r) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  if (b == 0) { c = hd_sgn(a); s = 0; r = fabs(a); }
  else if (a == 0) { c = 0; s = hd_sgn(b); r = fabs(b); }
  else if (fabs(b) > fabs(a)) {
    const double tau = a / b;
    s = hd_sgn(b) / sqrt(1 + tau * tau);
    c = s * tau;
    r = b / s;
  } else {
    const double tau = b / a;
    c = hd_sgn(a) / sqrt(1 + tau * tau);
    s = c * tau;
    r = a / c;
  }
}
// after alpha and beta of this iteration are known (s.alpha, s.beta): rotations, the three vector-update coefficients,
// norm estimates (lsmr.py:379-440)
MVUS_SOLVER_HD void lsmr_rotations(LsmrScalars& s) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double alpha = s.alpha, beta = s.beta;
  double chat, shat, alphahat; hd_sym_ortho(s.alphabar, s.damp, chat, shat, alphahat);
  const double rhoold = s.rho;
  double c, sn; hd_sym_ortho(alphahat, beta, c, sn, s.rho);
  const double thetanew = sn * alpha;
  s.alphabar = c * alpha;
  const double rhobarold = s.rhobar, zetaold = s.zeta;
  const double thetabar = s.sbar * s.rho, rhotemp = s.cbar * s.rho;
  hd_sym_ortho(s.cbar * s.rho, thetanew, s.cbar, s.sbar, s.rhobar);
  s.zeta = s.cbar * s.zetabar;
  s.zetabar = -s.sbar * s.zetabar;
  s.c_hbar = -(thetabar * s.rho / (rhoold * rhobarold));
  s.c_x = s.zeta / (s.rho * s.rhobar);
  s.c_h = -(thetanew / s.rho);
  const double betaacute = chat * s.betadd, betacheck = -shat * s.betadd;
  const double betahat = c * betaacute;
  s.betadd = -sn * betaacute;
  const double thetatildeold = s.thetatilde;
  double ctildeold, stildeold, rhotildeold; hd_sym_ortho(s.rhodold, thetabar, ctildeold, stildeold, rhotildeold);
  s.thetatilde = stildeold * s.rhobar;
  s.rhodold = ctildeold * s.rhobar;
  s.betad = -stildeold * s.betad + ctildeold * betahat;
  s.tautildeold = (zetaold - thetatildeold * s.tautildeold) / rhotildeold;
  const double taud = (s.zeta - s.thetatilde * s.tautildeold) / s.rhodold;
  s.d = s.d + betacheck * betacheck;
  s.normr = sqrt(s.d + (s.betad - taud) * (s.betad - taud) + s.betadd * s.betadd);
  s.normA2 = s.normA2 + beta * beta;
  s.normA = sqrt(s.normA2);
  s.normA2 = s.normA2 + alpha * alpha;
  s.maxrbar = s.maxrbar > rhobarold ? s.maxrbar : rhobarold;
  if (s.itn > 1) s.minrbar = s.minrbar < rhobarold ? s.minrbar : rhobarold;
  s.condA = (s.maxrbar > rhotemp ? s.maxrbar : rhotemp) / (s.minrbar < rhotemp ? s.minrbar : rhotemp);
  s.normar = fabs(s.zetabar);
}
// the stopping tests once |x| is known (lsmr.py:442-470)
MVUS_SOLVER_HD void lsmr_tests(LsmrScalars& s, double normx) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double test1 = s.normr / s.normb;
  const double test2 = (s.normA * s.normr) != 0 ? s.normar / (s.normA * s.normr) : 1.0 / 0.0;
  const double test3 = 1 / s.condA;
  const double t1 = test1 / (1 + s.normA * normx / s.normb);
  const double rtol = s.btol + s.atol * s.normA * normx / s.normb;
  int istop = 0;
  if (s.itn >= s.maxiter) istop = 7;
  if (1 + test3 <= 1) istop = 6;
  if (1 + test2 <= 1) istop = 5;
  if (1 + t1 <= 1) istop = 4;
  if (test3 <= s.ctol) istop = 3;
  if (test2 <= s.atol) istop = 2;
  if (test1 <= rtol) istop = 1;
  s.istop = istop;
}
}  // namespace detail

// ------------------------------------------------------------------------------------------------
// LSMR on A = [J diag(D); diag(E)] with right-hand side [b; 0] and scalar damping `damp`.
// D and E are device n-vectors or nullptr (D = 1, no extra rows).  Result in x_dev (n).
// ------------------------------------------------------------------------------------------------
template <class B>
struct Lsmr {
  B& be;
  int64_t n, m;
  double *ut, *ub, *v, *h, *hbar, *tn, *tm;
  explicit Lsmr(B& b) : be(b), n(b.n()), m(b.m_local()) {
    ut = be.alloc(m); tm = be.alloc(m);
    ub = be.alloc(n); v = be.alloc(n); h = be.alloc(n); hbar = be.alloc(n); tn = be.alloc(n);
  }
  ~Lsmr() { for (double* p : {ut, tm, ub, v, h, hbar, tn}) be.release(p); }

  // out = A^T [ut; ub]
  void rmatvec(const double* D, const double* E, const double* in_t, const double* in_b, double* out) {
    be.jtu(in_t, out);
    if (D) be.mul(n, D, out, out);
    if (E) { be.mul(n, E, in_b, tn); be.axpby(n, 1.0, out, 1.0, tn, out); }
  }

  int run(const double* D, const double* E, const double* b_top, double damp, double atol, double btol, double conlim,
          int64_t maxiter, double* x, int* itn_out) {
    using detail::sym_ortho;
    const bool trace = std::getenv("MVUS_LSMR_TRACE") != nullptr;
    be.copy(ut, b_top, m);
    if (E) be.fill(ub, 0.0, n);
    const double normb = std::sqrt(be.dot_m(ut, ut));
    double beta = normb, alpha = 0;
    be.fill(x, 0.0, n);
    if (beta > 0) {
      be.axpby(m, 1.0 / beta, ut, 0.0, ut, ut);
      rmatvec(D, E, ut, ub, v);
      alpha = std::sqrt(be.dot_n(v, v, n));
    } else {
      be.fill(v, 0.0, n);
    }
    if (alpha > 0) be.axpby(n, 1.0 / alpha, v, 0.0, v, v);
    LsmrScalars sc;
    sc.alpha = alpha; sc.beta = beta; sc.zetabar = alpha * beta; sc.alphabar = alpha;
    be.copy(h, v, n);
    be.fill(hbar, 0.0, n);
    sc.betadd = beta; sc.normA2 = alpha * alpha; sc.normb = normb; sc.damp = damp; sc.atol = atol; sc.btol = btol;
    sc.ctol = conlim > 0 ? 1 / conlim : 0;
    sc.normr = beta; sc.normar = alpha * beta; sc.normA = std::sqrt(sc.normA2); sc.maxiter = maxiter;
    if (itn_out) *itn_out = 0;
    if (sc.normar == 0) return 0;
    if (normb == 0) { be.fill(x, 0.0, n); return 0; }
    if constexpr (B::kDeviceLsmr) {
      // the HIP backend keeps the scalars on the device and runs batches of iterations without a host round trip
      // (three synchronisations and ~17 launches per iteration otherwise); same kernels' arithmetic, same bits
      if (!trace && be.lsmr_on_device() && (!(D || E) || be.lsmr_scaled_on_device())) {
        be.lsmr_iterations(sc, ut, tm, v, tn, h, hbar, x, D, E, ub);
        if (itn_out) *itn_out = (int)sc.itn;
        return sc.istop;
      }
    }
    while (sc.itn < maxiter) {
      ++sc.itn;
      // u = A v - alpha u ; beta = |u|
      if (D) { be.mul(n, D, v, tn); be.jv(tn, tm); } else be.jv(v, tm);
      be.axpby(m, 1.0, tm, -sc.alpha, ut, ut);
      double bsq = be.dot_m(ut, ut);
      if (E) {
        be.mul(n, E, v, tn);
        be.axpby(n, 1.0, tn, -sc.alpha, ub, ub);
        bsq += be.dot_n(ub, ub, n);
      }
      sc.beta = std::sqrt(bsq);
      if (sc.beta > 0) {
        be.axpby(m, 1.0 / sc.beta, ut, 0.0, ut, ut);
        if (E) be.axpby(n, 1.0 / sc.beta, ub, 0.0, ub, ub);
        // v = A^T u - beta v
        be.jtu(ut, tn);
        if (D) be.mul(n, D, tn, tn);
        be.axpby(n, 1.0, tn, -sc.beta, v, v);
        if (E) { be.mul(n, E, ub, tn); be.axpby(n, 1.0, v, 1.0, tn, v); }
        sc.alpha = std::sqrt(be.dot_n(v, v, n));
        if (sc.alpha > 0) be.axpby(n, 1.0 / sc.alpha, v, 0.0, v, v);
      }
      detail::lsmr_rotations(sc);
      // hbar = h - (thetabar*rho/(rhoold*rhobarold)) hbar ; x += (zeta/(rho*rhobar)) hbar ; h = v - (thetanew/rho) h
      be.axpby(n, 1.0, h, sc.c_hbar, hbar, hbar);
      be.axpby(n, 1.0, x, sc.c_x, hbar, x);
      be.axpby(n, 1.0, v, sc.c_h, h, h);
      const double normx = std::sqrt(be.dot_n(x, x, n));
      detail::lsmr_tests(sc, normx);
      if (trace) std::fprintf(stderr, "lsmr %3d alpha=%.10e beta=%.10e normr=%.6e normar=%.6e normA=%.4e condA=%.4e\n", (int)sc.itn, sc.alpha, sc.beta, sc.normr, sc.normar, sc.normA, sc.condA);
      if (sc.istop > 0) break;
    }
    const int64_t itn = sc.itn;
    const int istop = sc.istop;
    if (itn_out) *itn_out = (int)itn;
    return istop;
  }
};

// ------------------------------------------------------------------------------------------------
// Trust Region Reflective driver (scipy trf_no_bounds / trf_bounds with tr_solver='lsmr').
// x: in/out host vector; lb/ub: host bounds (all +-inf -> unbounded variant, as scipy's trf()).
// On return the backend's f buffer `f_dev` holds f(x).
// ------------------------------------------------------------------------------------------------
template <class B>
SolveResult trf_lsmr(B& be, std::vector<double>& x, const std::vector<double>& lb, const std::vector<double>& ub,
                     const SolveOptions& opt, double* f_dev) {
  using namespace detail;
  SolveResult res;
  const int64_t n = be.n(), m = be.m_local();
  bool bounded = false;
  for (int64_t i = 0; i < n; ++i) if (lb[i] != -std::numeric_limits<double>::infinity() || ub[i] != std::numeric_limits<double>::infinity()) bounded = true;
  if (!in_bounds(x, lb, ub)) { res.error = -3; return res; }
  if (bounded) make_strictly_feasible(x, lb, ub, 1e-10);

  PoolGuard<B> pool(be);
  double* x_dev = pool.get(n);
  double* xt_dev = pool.get(n);
  double* f_new = pool.get(m);
  double* JS0 = pool.get(m);
  double* JS1 = pool.get(m);
  double* tmp_m = pool.get(m);
  double* tmp_m2 = pool.get(m);
  double* vec_n = pool.get(n);
  double* D_dev = pool.get(n);
  double* E_dev = pool.get(n);
  double* gn_dev = pool.get(n);
  Lsmr<B> lsmr(be);
  auto cleanup = [] {};      // the pool guard returns the buffers

  std::vector<double> g(n), v, dv, d(n, 1.0), diag_h(n, 0.0), g_h(n), gn_h(n), S0(n), S1(n), tmpn(n);
  be.upload(x_dev, x.data(), n);
  be.jacobian(x_dev, f_dev, opt.jac_mode);
  res.nfev = 1; res.njev = 1;
  double cost = 0.5 * be.dot_m(f_dev, f_dev);
  if (!std::isfinite(cost)) { res.error = -3; cleanup(); return res; }
  res.initial_cost = cost;
  be.jtu(f_dev, vec_n);
  be.download(g.data(), vec_n, n);

  double Delta;
  if (bounded) {
    CL_scaling_vector(x, g, lb, ub, v, dv);
    double s = 0; for (int64_t i = 0; i < n; ++i) { const double q = x[i] / std::sqrt(v[i]); s += q * q; }
    Delta = std::sqrt(s);
  } else {
    Delta = norm2(x);
  }
  if (Delta == 0) Delta = 1.0;

  const int64_t lsmr_maxiter = opt.lsmr_maxiter > 0 ? opt.lsmr_maxiter : std::min<int64_t>(be.m_global() + (bounded ? n : 0), n);
  int status = -1;
  double g_norm = 0, step_norm = 0, actual_reduction = 0;
  std::vector<double> x_new(n), step(n), step_h(n), p_h(n), p(n), r_h(n), r(n), ag_h(n), ag(n);

  // helper: y_dev[m] = J_h s = J (d * s)
  auto Jh = [&](const std::vector<double>& s, double* y_dev) {
    for (int64_t i = 0; i < n; ++i) tmpn[i] = d[i] * s[i];
    be.upload(vec_n, tmpn.data(), n);
    be.jv(vec_n, y_dev);
  };

  while (true) {
    if (bounded) {
      CL_scaling_vector(x, g, lb, ub, v, dv);
      g_norm = 0; for (int64_t i = 0; i < n; ++i) g_norm = std::max(g_norm, std::fabs(g[i] * v[i]));
    } else {
      g_norm = 0; for (int64_t i = 0; i < n; ++i) g_norm = std::max(g_norm, std::fabs(g[i]));
    }
    if (g_norm < opt.gtol) status = 1;
    if (status != -1 || res.nfev == opt.max_nfev) break;

    if (bounded) {
      for (int64_t i = 0; i < n; ++i) { d[i] = std::sqrt(v[i]); diag_h[i] = g[i] * dv[i]; g_h[i] = d[i] * g[i]; }
    } else {
      for (int64_t i = 0; i < n; ++i) { d[i] = 1.0; diag_h[i] = 0.0; g_h[i] = g[i]; }
    }
    // regularisation term from the 1-D Cauchy model (trf.py:306-310 / :460-464)
    const double gh_norm = norm2(g_h);
    Jh(g_h, tmp_m);
    double a = be.dot_m(tmp_m, tmp_m);
    if (bounded) for (int64_t i = 0; i < n; ++i) a += g_h[i] * diag_h[i] * g_h[i];
    a *= 0.5;
    const double b = -gh_norm * gh_norm;
    double tq, ag_value;
    minimize_quadratic_1d(a, b, 0.0, Delta / gh_norm, 0.0, tq, ag_value);
    const double reg_term = -ag_value / (Delta * Delta);

    int itn = 0;
    if (bounded) {
      for (int64_t i = 0; i < n; ++i) tmpn[i] = std::sqrt(diag_h[i] + reg_term);
      be.upload(E_dev, tmpn.data(), n);
      be.upload(D_dev, d.data(), n);
      lsmr.run(D_dev, E_dev, f_dev, 0.0, opt.lsmr_atol, opt.lsmr_btol, opt.lsmr_conlim, lsmr_maxiter, gn_dev, &itn);
    } else {
      lsmr.run(nullptr, nullptr, f_dev, std::sqrt(reg_term), opt.lsmr_atol, opt.lsmr_btol, opt.lsmr_conlim, lsmr_maxiter, gn_dev, &itn);
    }
    res.lin_iters += itn;
    be.download(gn_h.data(), gn_dev, n);
    // S = qr([g_h, gn_h]) (economic): orthonormal basis of the 2-D subspace
    for (int64_t i = 0; i < n; ++i) S0[i] = g_h[i] / gh_norm;
    {
      double pr = dot(S0, gn_h);
      for (int64_t i = 0; i < n; ++i) S1[i] = gn_h[i] - pr * S0[i];
      pr = dot(S0, S1);
      for (int64_t i = 0; i < n; ++i) S1[i] -= pr * S0[i];
      const double nn = norm2(S1);
      if (nn > 1e-300 * (1.0 + norm2(gn_h))) for (int64_t i = 0; i < n; ++i) S1[i] /= nn;
      else std::fill(S1.begin(), S1.end(), 0.0);
    }
    Jh(S0, JS0);
    Jh(S1, JS1);
    double BJ[4];   // JS^T JS
    BJ[0] = be.dot_m(JS0, JS0); BJ[1] = BJ[2] = be.dot_m(JS0, JS1); BJ[3] = be.dot_m(JS1, JS1);
    double BS[4] = {BJ[0], BJ[1], BJ[2], BJ[3]};
    if (bounded) {
      double s00 = 0, s01 = 0, s11 = 0;
      for (int64_t i = 0; i < n; ++i) { s00 += S0[i] * diag_h[i] * S0[i]; s01 += S0[i] * diag_h[i] * S1[i]; s11 += S1[i] * diag_h[i] * S1[i]; }
      BS[0] += s00; BS[1] += s01; BS[2] += s01; BS[3] += s11;
    }
    if (BS[3] == 0 && norm2(S1) == 0) BS[3] = 1.0;  // degenerate 1-D subspace
    const double gS[2] = {dot(S0, g_h), dot(S1, g_h)};
    const double theta = std::max(0.995, 1 - g_norm);
    // value of the quadratic model at a subspace point q (coordinates in S)
    auto quad_S = [&](const double q[2]) {
      return 0.5 * (q[0] * (BS[0] * q[0] + BS[1] * q[1]) + q[1] * (BS[2] * q[0] + BS[3] * q[1])) + gS[0] * q[0] + gS[1] * q[1];
    };

    if (opt.verbose >= 2) std::fprintf(stderr, "trf: cost=%.12e Delta=%.12e reg=%.12e lsmr_itn=%d BS=[%.12e %.12e %.12e] gS=[%.12e %.12e]\n", cost, Delta, reg_term, itn, BS[0], BS[1], BS[3], gS[0], gS[1]);
    actual_reduction = -1;
    double cost_new = cost, predicted_reduction = 0;
    while (actual_reduction <= 0 && res.nfev < opt.max_nfev) {
      double pS[2];
      solve_trust_region_2d(BS, gS, Delta, pS);
      for (int64_t i = 0; i < n; ++i) p_h[i] = S0[i] * pS[0] + S1[i] * pS[1];
      if (!bounded) {
        step_h = p_h;
        predicted_reduction = -quad_S(pS);
        for (int64_t i = 0; i < n; ++i) { step[i] = step_h[i]; x_new[i] = x[i] + step[i]; }
      } else {
        // select_step (trf.py:128-202)
        for (int64_t i = 0; i < n; ++i) { p[i] = d[i] * p_h[i]; x_new[i] = x[i] + p[i]; }
        if (in_bounds(x_new, lb, ub)) {
          step = p; step_h = p_h; predicted_reduction = -quad_S(pS);
        } else {
          std::vector<int> hits;
          const double p_stride = step_size_to_bound(x, p, lb, ub, &hits);
          for (int64_t i = 0; i < n; ++i) { r_h[i] = hits[i] ? -p_h[i] : p_h[i]; r[i] = d[i] * r_h[i]; }
          for (int64_t i = 0; i < n; ++i) { p[i] *= p_stride; p_h[i] *= p_stride; }
          double pSs[2] = {pS[0] * p_stride, pS[1] * p_stride};
          std::vector<double> x_on_bound(n);
          for (int64_t i = 0; i < n; ++i) x_on_bound[i] = x[i] + p[i];
          double tneg, to_tr;
          intersect_trust_region(p_h, r_h, Delta, tneg, to_tr);
          const double to_bound = step_size_to_bound(x_on_bound, r, lb, ub, nullptr);
          double r_stride = std::min(to_bound, to_tr), r_stride_l, r_stride_u;
          if (r_stride > 0) {
            r_stride_l = (1 - theta) * p_stride / r_stride;
            r_stride_u = (r_stride == to_bound) ? theta * to_bound : to_tr;
          } else { r_stride_l = 0; r_stride_u = -1; }
          double r_value = std::numeric_limits<double>::infinity();
          if (r_stride_l <= r_stride_u) {
            // build_quadratic_1d(J_h, g_h, r_h, s0=p_h, diag=diag_h)
            Jh(r_h, tmp_m);                                         // v = J_h r_h
            be.axpby(m, pSs[0], JS0, pSs[1], JS1, tmp_m2);          // u = J_h p_h
            double qa = be.dot_m(tmp_m, tmp_m), qb = dot(g_h, r_h) + be.dot_m(tmp_m2, tmp_m);
            double qc = 0.5 * be.dot_m(tmp_m2, tmp_m2) + dot(g_h, p_h);
            for (int64_t i = 0; i < n; ++i) { qa += r_h[i] * diag_h[i] * r_h[i]; qb += p_h[i] * diag_h[i] * r_h[i]; qc += 0.5 * p_h[i] * diag_h[i] * p_h[i]; }
            qa *= 0.5;
            double rs_;
            minimize_quadratic_1d(qa, qb, r_stride_l, r_stride_u, qc, rs_, r_value);
            for (int64_t i = 0; i < n; ++i) { r_h[i] = r_h[i] * rs_ + p_h[i]; r[i] = r_h[i] * d[i]; }
          }
          for (int64_t i = 0; i < n; ++i) { p[i] *= theta; p_h[i] *= theta; }
          pSs[0] *= theta; pSs[1] *= theta;
          const double p_value = quad_S(pSs);
          for (int64_t i = 0; i < n; ++i) { ag_h[i] = -g_h[i]; ag[i] = d[i] * ag_h[i]; }
          double to_tr2 = Delta / norm2(ag_h);
          const double to_bound2 = step_size_to_bound(x, ag, lb, ub, nullptr);
          double ag_stride = (to_bound2 < to_tr2) ? theta * to_bound2 : to_tr2;
          // build_quadratic_1d(J_h, g_h, -g_h, diag_h) is the (a, b) of the regularisation step above
          double ag_val;
          { const double ub_ = ag_stride; minimize_quadratic_1d(a, b, 0.0, ub_, 0.0, ag_stride, ag_val); }
          for (int64_t i = 0; i < n; ++i) { ag_h[i] *= ag_stride; ag[i] *= ag_stride; }
          if (p_value < r_value && p_value < ag_val) { step = p; step_h = p_h; predicted_reduction = -p_value; }
          else if (r_value < p_value && r_value < ag_val) { step = r; step_h = r_h; predicted_reduction = -r_value; }
          else { step = ag; step_h = ag_h; predicted_reduction = -ag_val; }
        }
        for (int64_t i = 0; i < n; ++i) x_new[i] = x[i] + step[i];
        make_strictly_feasible(x_new, lb, ub, 0.0);
      }
      be.upload(xt_dev, x_new.data(), n);
      be.residual(xt_dev, f_new);
      ++res.nfev;
      const double step_h_norm = norm2(step_h);
      cost_new = 0.5 * be.dot_m(f_new, f_new);
      if (!std::isfinite(cost_new)) { Delta = 0.25 * step_h_norm; continue; }
      actual_reduction = cost - cost_new;
      double ratio, Delta_new = Delta;
      update_tr_radius(Delta_new, actual_reduction, predicted_reduction, step_h_norm, step_h_norm > 0.95 * Delta, ratio);
      step_norm = norm2(step);
      const int term = check_termination(actual_reduction, cost, step_norm, norm2(x), ratio, opt.ftol, opt.xtol);
      if (term != -1) { status = term; break; }
      Delta = Delta_new;
    }
    if (actual_reduction > 0) {
      x = x_new;
      cost = cost_new;
      be.copy(x_dev, xt_dev, n);
      be.jacobian(x_dev, f_dev, opt.jac_mode);   // recomputes f(x) as well
      ++res.njev;
      be.jtu(f_dev, vec_n);
      be.download(g.data(), vec_n, n);
    } else {
      step_norm = 0; actual_reduction = 0;
    }
  }
  if (status == -1) status = 0;
  res.status = status;
  res.cost = cost;
  res.optimality = g_norm;
  cleanup();
  return res;
}

}  // namespace mvus
