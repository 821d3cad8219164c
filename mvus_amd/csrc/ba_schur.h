// Levenberg-Marquardt on the Gauss-Newton normal equations with a Schur split between the small
// camera/sync block and the block-banded spline block (north star of BASELINE.json).
//
//     H = J^T J = [ A   E ]      A: block diagonal, one (3+P)x(3+P) block per camera (alpha,beta,rs,pose[,K,d])
//                 [ E^T C ]      C: spline block, block-banded with 3x3 blocks (cubic B-spline: 4 active control
//     g = J^T f                     points per row; motion rows widen the band), E: camera x spline cross block
//
//     (H + lambda D) p = -g,  D = diag(H) (Marquardt scaling), solved by eliminating the spline block:
//         Z   = (C + lambda D_s)^-1 [E^T | g_s]               block-banded Cholesky, many right-hand sides
//         S   = (A + lambda D_c) - E Z_E                        dense reduced camera system, <= 64*18 unknowns
//         p_c = -S^-1 (g_c - E z_g),   p_s = -(z_g + Z_E p_c)
//
// The reference has no counterpart (scipy's trf/lsmr never forms H); the solver exists because assembling and
// factorising H on the GPU is far cheaper than ~100 LSMR passes over J per step.  Termination tests, evaluation
// counting and the result fields follow scipy's conventions (check_termination, nfev/njev/status) so that
// Scene.BA can switch solvers without changing its contract.
//
// `Schur` concept:  void assemble(B&, const double* f_dev);  void gradient(std::vector<double>& g);
//                   void diagonal(std::vector<double>& d);   bool solve(double lambda, std::vector<double>& p);
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <vector>

#include "ba_solver.h"

namespace mvus {

template <class B, class Schur>
SolveResult lm_schur(B& be, Schur& sc, std::vector<double>& x, const std::vector<double>& lb, const std::vector<double>& ub,
                     const SolveOptions& opt, double* f_dev) {
  using namespace detail;
  SolveResult res;
  const int64_t n = be.n(), m = be.m_local();
  if (!in_bounds(x, lb, ub)) { res.error = -3; return res; }
  double* x_dev = be.alloc(n);
  double* xt_dev = be.alloc(n);
  double* f_new = be.alloc(m);
  auto cleanup = [&]() { be.release(x_dev); be.release(xt_dev); be.release(f_new); };

  be.upload(x_dev, x.data(), n);
  be.jacobian(x_dev, f_dev, opt.jac_mode);
  res.nfev = 1; res.njev = 1;
  double cost = 0.5 * be.dot_m(f_dev, f_dev);
  if (!std::isfinite(cost)) { res.error = -3; cleanup(); return res; }
  res.initial_cost = cost;
  sc.assemble(be, f_dev);
  std::vector<double> g(n), D(n), p(n), x_new(n), step(n);
  sc.gradient(g);
  sc.diagonal(D);

  double lambda = opt.lm_lambda0 > 0 ? opt.lm_lambda0 : 1e-4, nu = 2.0;
  int status = -1;
  double g_norm = 0;
  while (true) {
    g_norm = 0;
    for (int64_t i = 0; i < n; ++i) {
      // projected gradient: a component pushing against an active bound does not count
      const bool blocked = (x[i] <= lb[i] && g[i] > 0) || (x[i] >= ub[i] && g[i] < 0);
      if (!blocked) g_norm = std::max(g_norm, std::fabs(g[i]));
    }
    if (g_norm < opt.gtol) status = 1;
    if (status != -1 || res.nfev >= opt.max_nfev) break;

    double actual_reduction = -1, cost_new = cost;
    while (actual_reduction <= 0 && res.nfev < opt.max_nfev) {
      if (!sc.solve(lambda, p)) {           // not positive definite at this damping: raise it
        lambda *= 10.0;
        if (lambda > 1e12) { status = 0; break; }
        continue;
      }
      ++res.lin_iters;
      for (int64_t i = 0; i < n; ++i) {
        x_new[i] = std::min(std::max(x[i] + p[i], lb[i]), ub[i]);     // projection onto the rs box
        step[i] = x_new[i] - x[i];
      }
      // predicted reduction of the quadratic model: -(g.p + 0.5 p^T H p) = 0.5 (lambda p^T D p - g.p)
      double gp = 0, pDp = 0;
      for (int64_t i = 0; i < n; ++i) { gp += g[i] * step[i]; pDp += step[i] * D[i] * step[i]; }
      const double predicted = 0.5 * (lambda * pDp - gp);
      be.upload(xt_dev, x_new.data(), n);
      be.residual(xt_dev, f_new);
      ++res.nfev;
      cost_new = 0.5 * be.dot_m(f_new, f_new);
      if (!std::isfinite(cost_new)) { lambda *= nu; nu *= 2.0; continue; }
      actual_reduction = cost - cost_new;
      const double ratio = predicted > 0 ? actual_reduction / predicted : (actual_reduction > 0 ? 1.0 : 0.0);
      const double step_norm = norm2(step);
      const int term = check_termination(actual_reduction, cost, step_norm, norm2(x), ratio, opt.ftol, opt.xtol);
      if (opt.verbose >= 2)
        std::fprintf(stderr, "lm: nfev=%d cost=%.10e -> %.10e lambda=%.3e ratio=%.3f |step|=%.3e\n", res.nfev, cost, cost_new, lambda, ratio, step_norm);
      if (actual_reduction > 0) {
        const double t = 2.0 * ratio - 1.0;
        lambda *= std::max(1.0 / 3.0, 1.0 - t * t * t);                 // Nielsen's update
        nu = 2.0;
      } else {
        lambda *= nu; nu *= 2.0;
      }
      if (term != -1) { status = term; break; }
    }
    if (actual_reduction > 0) {
      x = x_new;
      cost = cost_new;
      be.copy(x_dev, xt_dev, n);
      if (status == -1 && res.nfev >= opt.max_nfev) {
        // evaluation budget spent: no further step will be taken, so the accepted point is not re-linearised (one
        // Jacobian + assembly saved per call); the reported optimality is then that of the last linearisation
        be.copy(f_dev, f_new, m);
        break;
      }
      be.jacobian(x_dev, f_dev, opt.jac_mode);
      ++res.njev;
      sc.assemble(be, f_dev);
      sc.gradient(g);
      sc.diagonal(D);
    }
    if (status != -1) continue;   // re-evaluate g_norm once, then leave through the break above
  }
  if (status == -1) status = 0;
  res.status = status;
  res.cost = cost;
  res.optimality = g_norm;
  res.lm_lambda = lambda;
  cleanup();
  return res;
}

// ------------------------------------------------------------------------------------------------
// Dense reference implementation of the Schur concept for the host test backend: builds H from the
// backend's J v operator and factorises (H + lambda diag(H)) with a dense Cholesky.  Test infrastructure.
// ------------------------------------------------------------------------------------------------
struct HostSchur {
  int64_t n = 0;
  std::vector<double> H, g;
  template <class B>
  void assemble(B& be, const double* f) {
    n = be.n();
    const int64_t m = be.m_local();
    std::vector<double> Jd((size_t)m * n), e(n, 0.0), col(m);
    for (int64_t j = 0; j < n; ++j) {
      e[j] = 1.0; be.jv(e.data(), col.data()); e[j] = 0.0;
      for (int64_t i = 0; i < m; ++i) Jd[(size_t)i * n + j] = col[i];
    }
    H.assign((size_t)n * n, 0.0); g.assign(n, 0.0);
    for (int64_t i = 0; i < m; ++i) {
      const double* r = &Jd[(size_t)i * n];
      std::vector<int> nz;
      for (int64_t j = 0; j < n; ++j) if (r[j] != 0.0) nz.push_back((int)j);
      for (int a : nz) { g[a] += r[a] * f[i]; for (int b : nz) H[(size_t)a * n + b] += r[a] * r[b]; }
    }
  }
  void gradient(std::vector<double>& out) { out = g; }
  void diagonal(std::vector<double>& d) { d.resize(n); for (int64_t i = 0; i < n; ++i) d[i] = damp_scale(H[(size_t)i * n + i]); }
  static double damp_scale(double hii) { return hii > 0 ? hii : 1.0; }
  bool solve(double lambda, std::vector<double>& p) {
    std::vector<double> L(H);
    for (int64_t i = 0; i < n; ++i) L[(size_t)i * n + i] += lambda * damp_scale(H[(size_t)i * n + i]);
    for (int64_t k = 0; k < n; ++k) {
      double d = L[(size_t)k * n + k];
      for (int64_t j = 0; j < k; ++j) d -= L[(size_t)k * n + j] * L[(size_t)k * n + j];
      if (!(d > 0)) return false;
      d = std::sqrt(d);
      L[(size_t)k * n + k] = d;
      for (int64_t i = k + 1; i < n; ++i) {
        double s = L[(size_t)i * n + k];
        for (int64_t j = 0; j < k; ++j) s -= L[(size_t)i * n + j] * L[(size_t)k * n + j];
        L[(size_t)i * n + k] = s / d;
      }
    }
    p.assign(n, 0.0);
    for (int64_t i = 0; i < n; ++i) {
      double s = -g[i];
      for (int64_t j = 0; j < i; ++j) s -= L[(size_t)i * n + j] * p[j];
      p[i] = s / L[(size_t)i * n + i];
    }
    for (int64_t i = n - 1; i >= 0; --i) {
      double s = p[i];
      for (int64_t j = i + 1; j < n; ++j) s -= L[(size_t)j * n + i] * p[j];
      p[i] = s / L[(size_t)i * n + i];
    }
    return true;
  }
};

}  // namespace mvus
