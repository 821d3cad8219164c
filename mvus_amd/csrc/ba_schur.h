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
// The iteration is driven from the host but keeps every vector in the backend's memory: one LM iteration is ONE
// batch of asynchronous work -- projected-gradient norm, damped solve, trial point with the model's predicted
// reduction, residuals at the trial point, their squared norm -- followed by ONE fetch of seven scalars on which
// the host decides (accept / reject, damping update, termination).  The trial is launched speculatively before the
// gradient-norm test is known; when that test terminates the solve the trial is simply dropped (not counted).
//
// `Schur` concept:  void linearize(B&, x, f, jac_mode, f_valid);    Jacobian at x (f = f(x) is already there when f_valid)
//                                                                 and the normal equations; may fuse the two
//                   void assemble(B&, const double* f);            normal equations of the backend's Jacobian
//                   const double* grad_ptr(), diag_ptr();          g = J^T f and D = diag(H) (1 where zero), x order
//                   void solve_async(double lambda);               p = -(H + lambda D)^-1 g -> step_ptr()
//                   const double* step_ptr(); const int* fail_ptr();  bool solve_ok();   (solve_ok after a fetch)
//                   bool retry_same();   after a failed solve: true = not a numerical failure, repeat it at the same lambda
//                   bool spec_ok(jac_mode); void linearize_spec(B&, x, f, jac_mode); void adopt_spec(); void drop_spec();   the linearisation at a TRIAL point,
//                                        enqueued into a second set of blocks before the host knows whether the trial is accepted
// Backend additions: set_bounds(lb, ub) -> lb_ptr()/ub_ptr(); lm_scalars() (>= 8 doubles); dot_m_into(a, b, out);
//                   lm_gnorm(x, lb, ub, g, out); lm_trial(x, p, lb, ub, g, D, fail, x_new, out4, gnorm_out) (the trial
//                   kernel reads x, g and the bounds anyway, so it also delivers the gradient norm); fetch(src, k, host).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <vector>

#include "ba_solver.h"

namespace mvus {

// host restatements of the two small vector kernels (used by the host test backend; the HIP backend has kernels)
// projected gradient: a component pushing against an active bound does not count
inline double lm_gnorm_host(int64_t n, const double* x, const double* lb, const double* ub, const double* g) {
  double gn = 0;
  for (int64_t i = 0; i < n; ++i) {
    const bool blocked = (x[i] <= lb[i] && g[i] > 0) || (x[i] >= ub[i] && g[i] < 0);
    if (!blocked) gn = std::max(gn, std::fabs(g[i]));
  }
  return gn;
}
// trial point x_new = P(x + p) (projection onto the rs box) and out = [g.step, step^T D step, |step|^2, |x|^2];
// a failed or non-finite solve gives step 0 and out[0] = NaN
inline void lm_trial_host(int64_t n, const double* x, const double* p, const double* lb, const double* ub, const double* g,
                          const double* D, int fail, double* x_new, double* out, double* gnorm_out, double cut = 1.0) {
  double gp = 0, pDp = 0, s2 = 0, x2 = 0;
  bool bad = fail != 0;
  for (int64_t i = 0; i < n && !bad; ++i) bad = !std::isfinite(p[i]);
  for (int64_t i = 0; i < n; ++i) {
    const double xn = bad ? x[i] : std::min(std::max(x[i] + cut * p[i], lb[i]), ub[i]);
    const double st = xn - x[i];
    x_new[i] = xn;
    gp += g[i] * st; pDp += st * D[i] * st; s2 += st * st; x2 += x[i] * x[i];
  }
  out[0] = bad ? std::numeric_limits<double>::quiet_NaN() : gp; out[1] = pDp; out[2] = s2; out[3] = x2;
  *gnorm_out = lm_gnorm_host(n, x, lb, ub, g);
}

template <class B, class Schur>
SolveResult lm_schur(B& be, Schur& sc, double* x, const std::vector<double>& lb, const std::vector<double>& ub,
                     const SolveOptions& opt, double* f_dev) {          // f_dev: the backend's residual buffer (f(x) on return)
  // x: the caller's buffer, n values, in and out -- read at the start, written once at the end (a vector in between was two 122 KB
  // copies and an allocation per call at configs[2], with the GPU idle: every microsecond of host work before the first launch and
  // after the last fetch is a microsecond of the step)
  using namespace detail;
  SolveResult res;
  const int64_t n = be.n(), m = be.m_local();
  PoolGuard<B> pool(be);             // returned on every exit, exceptions included (a time shard can throw out of solve_ok)
  // the current and the trial point live in two buffers the backend keeps from solve to solve: a caller that continues from the point
  // the previous solve returned (an outer loop of short solves -- bench.py's steps, Scene.BA after remove_outliers) finds it on the
  // device already (lm_resume: bitwise comparison with the host copy of that point) and x does not cross PCIe again
  int xcur = be.lm_resume(x);
  const bool resumed = xcur >= 0;
  // (a resumed point is one this driver returned: the caller's x0 of that solve, checked then, or a trial point, projected onto the
  // box by the trial kernel -- and the box is the handle's; the O(n) check would only delay the first launch)
  if (!resumed) for (int64_t i = 0; i < n; ++i) if (!(x[i] >= lb[i] && x[i] <= ub[i])) { res.error = -3; return res; }
  if (!resumed) xcur = 0;
  double* x_dev = be.lm_xbuf(xcur);
  double* xt_dev = be.lm_xbuf(xcur ^ 1);
  double* f_new = pool.get(m);
  auto cleanup = [] {};
  be.set_bounds(lb, ub);
  const double* lbp = be.lb_ptr();
  const double* ubp = be.ub_ptr();
  double* S = be.lm_scalars();   // [0] |f|^2 at x0, [1] projected |g|_inf, [2..5] trial scalars, [6] |f(x_trial)|^2
  double hs[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  if (!resumed) be.upload(x_dev, x, n);
  // A resumed point may come with more than its coordinates: the previous solve left f(x) in the residual buffer (its last accepted
  // trial's, or the start's) and, if its last act was a linearisation at x, the normal equations -- nothing else has touched the handle
  // since (lm_carry checks that and disarms itself).  The evaluation and the linearisation below are then the values already there.
  const LmCarry carry = resumed ? be.lm_carry(opt.jac_mode) : LmCarry{};
  double cost = 0;
  bool cost_known = false;
  // (the first linearisation follows at once: its storage is zeroed beside this evaluation where the backend can do that)
  if (carry.f_valid) { cost = carry.cost; res.initial_cost = cost; cost_known = true; }
  else if (be.residual_sq(x_dev, f_dev, S, sc.clear_ptr(), sc.clear_len())) sc.mark_cleared();
  res.nfev = 1; res.njev = 1;
  if (!(carry.f_valid && carry.lin_valid)) sc.linearize(be, x_dev, f_dev, opt.jac_mode, true);
  bool lin_at_x = true;                 // the blocks the solver holds are those of x_dev

  // f(x_new) becomes f(x): the two buffers change roles where the backend allows it (the HIP backend's are pool buffers of
  // one size), else a copy
  auto accept_residual = [&] {
    if (B::kSwapResiduals) { double* old = f_new; be.adopt_residual(f_dev, f_new); pool.replace(old, f_new); }
    else be.copy(f_dev, f_new, m);
  };
  int mir_cur = -1, mir_trial = 0;      // host mirrors of the accepted / the trial point (backend permitting)
  bool spec_live = false;               // the current trial's linearisation is enqueued (sc.linearize_spec)
  try {
  // Trust region in the reference's metric (mvus_solve_opts.lm_trust_radius): scipy's TRF with x_scale = 1 bounds |step|_2 by Delta
  // (scipy/optimize/_lsq/trf.py: Delta_0 = |x0|, or 1 when that is 0) and updates it with update_tr_radius (_lsq/common.py).  The
  // damped step p(lambda) is cut back to Delta along its direction by the trial kernel; S[7] brings |p|^2 back.
  const bool tr = opt.lm_trust_radius >= 0;
  double Delta = 0;
  if (tr) {
    Delta = opt.lm_trust_radius;
    if (!(Delta > 0)) { double s2 = 0; for (int64_t i = 0; i < n; ++i) s2 += x[i] * x[i]; Delta = s2 > 0 ? std::sqrt(s2) : 1.0; }
  }
  const int nfetch_all = sc.fail_in_scalars() ? 10 : (tr ? 8 : 7);      // (time shards: + the solve's failure flags, summed over the ranks)
  auto launch_trial = [&](double lambda) {
    sc.solve_async(lambda, true);
    be.lm_trial(x_dev, sc.step_ptr(), lbp, ubp, sc.grad_ptr(), sc.diag_ptr(), sc.fail_ptr(), xt_dev, S + 2, S + 1, be.mirror_dev(mir_trial),
                tr ? S + 7 : nullptr, Delta, sc.fail_sum_ptr());
    be.residual_sq(xt_dev, f_new, S + 6);
    // Most trials are accepted, and an accepted trial is followed by the linearisation at its point: that linearisation is enqueued NOW,
    // into the solver's second set of blocks, behind a marker the fetch below waits for instead of the whole stream -- the GPU works on
    // it while the host wakes up, decides and enqueues the next solve (20 - 60 us of idle device per iteration otherwise).  A rejected
    // trial leaves the blocks of x untouched (the next solve reads those) and the speculative set is simply written again.
    spec_live = false;
    if (sc.spec_ok(opt.jac_mode)) { be.fetch_enqueue(S, nfetch_all); sc.linearize_spec(be, xt_dev, f_new, opt.jac_mode); spec_live = true; }      // (marks where the fetch stops waiting)
  };
  const int nfetch = nfetch_all;

  double lambda = std::max(opt.lm_lambda0 > 0 ? opt.lm_lambda0 : 1e-4, opt.lm_lambda_min), nu = opt.lm_nu0 > 0 ? opt.lm_nu0 : 2.0;
  const double lambda_min = opt.lm_lambda_min;
  int status = -1;
  double g_norm = 0;
  while (true) {
    const bool can_try = res.nfev < opt.max_nfev;
    if (can_try) launch_trial(lambda);             // speculative: dropped if the gradient test below ends the solve
    else be.lm_gnorm(x_dev, lbp, ubp, sc.grad_ptr(), S + 1);
    be.fetch(S, can_try ? nfetch : 2, hs);
    if (!cost_known) {
      cost = 0.5 * hs[0];
      if (!std::isfinite(cost)) { res.error = -3; cleanup(); return res; }
      res.initial_cost = cost;
      cost_known = true;
    }
    g_norm = hs[1];
    if (g_norm < opt.gtol) status = 1;
    if (status != -1 || !can_try) break;

    double actual_reduction = -1, cost_new = cost;
    bool have_trial = true;
    while (actual_reduction <= 0 && res.nfev < opt.max_nfev) {
      if (!have_trial) { launch_trial(lambda); be.fetch(S + 2, nfetch - 2, hs + 2); }
      have_trial = false;
      if (!sc.solve_ok() || !std::isfinite(hs[2]) || !std::isfinite(hs[3])) {   // not positive definite at this damping: raise it
        if (spec_live) { sc.drop_spec(); spec_live = false; }
        if (sc.retry_same()) continue;      // (not a numerical failure -- an in-launch hand-over timed out: the same solve again, other route)
        lambda *= 10.0;
        if (lambda > 1e12) { status = 0; break; }
        continue;
      }
      ++res.lin_iters;
      // predicted reduction of the quadratic model: -(g.p + 0.5 p^T H p) = 0.5 (lambda p^T D p - g.p)
      // (a step cut back to the trust region, st = cut p: -g.st - st^T H st / 2 with p^T H p = -g.p - lambda p^T D p)
      const double cut = (tr && hs[7] > Delta * Delta) ? Delta / std::sqrt(hs[7]) : 1.0;
      const double predicted = cut < 1.0 ? 0.5 * lambda * hs[3] - (1.0 - 0.5 * cut) * hs[2] : 0.5 * (lambda * hs[3] - hs[2]);
      ++res.nfev;
      cost_new = 0.5 * hs[6];
      if (!std::isfinite(cost_new)) { if (spec_live) { sc.drop_spec(); spec_live = false; } lambda *= nu; nu *= 2.0; continue; }
      actual_reduction = cost - cost_new;
      const double ratio = predicted > 0 ? actual_reduction / predicted : (actual_reduction > 0 ? 1.0 : 0.0);
      const double step_norm = std::sqrt(hs[4]);
      const int term = check_termination(actual_reduction, cost, step_norm, std::sqrt(hs[5]), ratio, opt.ftol, opt.xtol);
      if (opt.verbose >= 2)
        std::fprintf(stderr, "lm: nfev=%d cost=%.10e -> %.10e lambda=%.3e ratio=%.3f |step|=%.3e\n", res.nfev, cost, cost_new, lambda, ratio, step_norm);
      const double lambda_used = lambda;
      if (actual_reduction > 0) {
        const double t = 2.0 * ratio - 1.0;
        lambda *= std::max(1.0 / 3.0, 1.0 - t * t * t);                 // Nielsen's update
        lambda = std::max(lambda, lambda_min);
        nu = 2.0;
      } else {
        lambda *= nu; nu *= 2.0;
        if (spec_live) { sc.drop_spec(); spec_live = false; }
      }
      if (tr) {                                                          // scipy's update_tr_radius
        const bool bound_hit = step_norm > 0.95 * Delta;
        if (ratio < 0.25) Delta = 0.25 * step_norm;
        else if (ratio > 0.75 && bound_hit) Delta *= 2.0;
        // a step that had to be cut to less than half: the next damped step should come out near the radius by itself (its
        // direction then turns from Gauss-Newton's towards the scaled gradient, as the exact trust-region step does)
        if (cut < 0.5) lambda = std::max(lambda, std::min(lambda_used * 0.5 / cut, lambda_used * 10.0));
        if (opt.verbose >= 2) std::fprintf(stderr, "lm:   trust region: |p|=%.3e cut=%.3f Delta -> %.3e\n", std::sqrt(hs[7]), cut, Delta);
      }
      if (term != -1) { status = term; break; }
    }
    if (actual_reduction > 0) {
      std::swap(x_dev, xt_dev);
      mir_cur = mir_trial; mir_trial ^= 1;
      cost = cost_new;
      if (status == -1 && res.nfev >= opt.max_nfev) {
        // evaluation budget spent: no further step will be taken, so the accepted point is not re-linearised (one
        // Jacobian + assembly saved per call); the reported optimality is then that of the last linearisation.
        // (With a speculative linearisation in flight the blocks of the accepted point exist anyway: they are kept for a caller that
        // continues from here -- lm_keep below -- and nobody waits for them now.)
        accept_residual();
        if (spec_live) { sc.adopt_spec(); res.async_tail = true; }
        else { lin_at_x = false; res.jac_stale = true; }   // J, span and the assembled blocks still belong to the previous point
        break;
      }
      accept_residual();                        // f at the accepted point is the trial residual: not evaluated again
      ++res.njev;
      if (spec_live) sc.adopt_spec();           // (enqueued before the fetch that decided: see launch_trial)
      else sc.linearize(be, x_dev, f_dev, opt.jac_mode, true);
    }
    if (status != -1) {                 // converged / gave up: report the gradient norm of the final linearisation
      be.lm_gnorm(x_dev, lbp, ubp, sc.grad_ptr(), S + 1);
      be.fetch(S + 1, 1, hs + 1);
      g_norm = hs[1];
      break;
    }
  }
  if (status == -1) status = 0;
  res.status = status;
  res.optimality = g_norm;
  res.lm_lambda = lambda;
  res.lm_nu = nu;
  } catch (...) {
    // a time shard whose rows have left the slice (backend: reshard_pending): not an error of the solve -- the point reached so far
    // goes back to the caller (error -4 -> MVUS_E_RESHARD), who re-cuts the timeline there and continues
    if (!be.reshard_pending()) throw;
    res.error = -4;
  }
  if (mir_cur < 0) { /* no step was accepted: x is the caller's x0 */ }
  else if (be.mirror_host(mir_cur)) std::copy(be.mirror_host(mir_cur), be.mirror_host(mir_cur) + n, x);   // written by the accepted trial's kernel, fetched since
  else be.download(x, x_dev, n);
  be.lm_remember(x_dev, x, mir_cur >= 0 ? be.mirror_host(mir_cur) : nullptr);
  if (!res.error) be.lm_keep(LmCarry{true, lin_at_x, cost}, opt.jac_mode);      // f_dev holds f(x_dev) on every regular exit
  res.cost = cost;
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
  void linearize(B& be, const double* x, double* f, int jac_mode, bool) { be.jacobian(x, f, jac_mode); assemble(be, f); }
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
    Dd.resize(n);
    for (int64_t i = 0; i < n; ++i) Dd[i] = damp_scale(H[(size_t)i * n + i]);
  }
  std::vector<double> Dd, pstep;
  int fail = 0;
  double* clear_ptr() { return nullptr; }
  int64_t clear_len() const { return 0; }
  void mark_cleared() {}
  const double* grad_ptr() const { return g.data(); }
  const double* diag_ptr() const { return Dd.data(); }
  const double* step_ptr() const { return pstep.data(); }
  const int* fail_ptr() const { return &fail; }
  bool solve_ok() const { return fail == 0; }
  bool retry_same() { return false; }
  bool spec_ok(int) const { return false; }
  template <class B> void linearize_spec(B&, const double*, double*, int) {}
  void adopt_spec() {}
  void drop_spec() {}
  void solve_async(double lambda, bool = false) { fail = solve(lambda, pstep) ? 0 : 1; if (fail) pstep.assign(n, 0.0); }
  bool fail_in_scalars() const { return false; }
  const double* fail_sum_ptr() const { return nullptr; }
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
