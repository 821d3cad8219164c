"""CPU checks of the restated scipy optimiser (mvus_amd/csrc/ba_solver.h) and of the block-sparse
Jacobian operator, run through the test-only host backend (tests/hostcheck)."""
import numpy as np
import pytest
from scipy import sparse
from scipy.optimize import least_squares

from oracle import ba_oracle as orc
from golden_util import CASES, load_case
from hostcheck_util import HostHandle
from mvus_amd import _lib
from mvus_amd import problem as mp


@pytest.mark.parametrize('name', CASES)
def test_full_residual_including_motion_rows(name):
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    h = HostHandle(prob)
    assert h.n == g['x0'].size and h.m == g['f_x0'].size
    for x, fref in ((g['x0'], g['f_x0']), (g['x0'] + g['delta'], g['f_x0_delta'])):
        f = h.residual(x)
        scale = np.maximum(1.0, np.abs(fref))
        assert np.max(np.abs(f - fref) / scale) < 1e-9
        assert np.array_equal(f == 0, fref == 0)


@pytest.mark.parametrize('name', CASES)
def test_dense_jacobian_vs_central_differences(name):
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    oprob, _ = orc.problem_from_scene(scene)
    h = HostHandle(prob)
    x = g['x0'] + g['delta']
    f, J = h.dense_jacobian(x, _lib.JAC_ANALYTIC)
    Jfd = orc.numeric_jacobian(oprob, x, rel=1e-6)
    if not prob.rs_free:
        Jfd[:, 2 * prob.C:3 * prob.C] = 0.0
    ok = np.abs(f) > 0.05
    if prob.motion_reg:                     # motion rows: sums of |.| over xyz, kink when a component is ~0
        ok[2 * prob.M:] = np.abs(f[2 * prob.M:]) > 1e-3 * np.abs(f[2 * prob.M:]).max()
    alpha, beta, rs, cams, tck = orc.unpack_x(oprob, x)
    near = []
    for c in range(prob.C):
        tau = orc.detection_to_global(oprob, c, alpha, beta, rs, cams[c])[0]
        ne = (np.abs(tau[:, None] - oprob.interval.reshape(1, -1)) < 0.05).any(axis=1)
        near += [ne, ne]
    ok[:2 * prob.M] &= ~np.concatenate(near)
    scale = np.maximum(np.abs(Jfd[ok]).max(axis=0), 1e-12)
    err = np.abs(J[ok] - Jfd[ok]) / scale
    if prob.motion_reg and prob.motion_type == 0:
        # F rows: |r_x|+|r_y|+|r_z| has kinks whenever one component crosses 0; tolerate those rows
        rowbad = (err > 2e-5).any(axis=1)
        assert rowbad[: int(ok[:2 * prob.M].sum())].sum() == 0
        assert rowbad.sum() <= 0.1 * max(1, ok[2 * prob.M:].sum())
    else:
        assert err.max() < 2e-5
    # adjoint consistency of the operator pair
    rng = np.random.default_rng(0)
    u = rng.normal(size=h.m)
    np.testing.assert_allclose(h.jtu(u), J.T @ u, rtol=1e-10, atol=1e-8)


def _scipy_vs_restatement(prob, x0, max_nfev, lsmr_maxiter):
    h = HostHandle(prob)
    h.prepare_pattern(x0)
    fun = lambda x: h.residual(x)
    jac = lambda x: sparse.csr_matrix(h.dense_jacobian(x, _lib.JAC_PATTERN)[1])
    lb, ub = prob.bounds()
    tr_options = dict(maxiter=lsmr_maxiter) if lsmr_maxiter else {}
    ref = least_squares(fun, x0, jac=jac, tr_solver='lsmr', tr_options=tr_options, xtol=1e-12, max_nfev=max_nfev,
                        bounds=(lb, ub) if prob.rs_bounds else (-np.inf, np.inf))
    opts = _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN, max_nfev)
    opts.lsmr_maxiter = lsmr_maxiter
    x, res, f = h.solve(x0, opts)
    return ref, x, res, f


# LSMR's Golub-Kahan recurrence on these Jacobians amplifies rounding by ~10x every 1-2 iterations
# (restatement vs scipy.sparse.linalg.lsmr on the same matrix: 1e-15 after 3 iterations, 8e-8 after 10,
# 2e-5 after 20), so two correct implementations that sum J v in a different order drift apart.  The
# trust-region logic is therefore pinned with LSMR capped at 4 iterations, where the whole 15-evaluation
# trajectory (accepted and rejected steps, Coleman-Li scaling) has to agree to ~1e-9.
@pytest.mark.parametrize('name', CASES)
def test_trf_restatement_matches_scipy_tight(name):
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    ref, x, res, f = _scipy_vs_restatement(prob, g['x0'], 15, 4)
    assert (res.nfev, res.njev, res.status) == (ref.nfev, ref.njev, ref.status)
    np.testing.assert_allclose(res.cost, ref.cost, rtol=1e-9)
    np.testing.assert_allclose(x, ref.x, rtol=0, atol=1e-7)
    np.testing.assert_allclose(f, ref.fun, rtol=0, atol=1e-6)
    np.testing.assert_allclose(res.optimality, ref.optimality, rtol=1e-6)


def test_trf_restatement_active_bounds():
    from mvus_amd import synth
    sc = synth.make_scene(3, 900, seed=5, rolling_shutter=True, rs_bounds=True, knot_spacing=14.0)
    sc.rs[:] = [0.999, 0.001, 0.5]
    prob, x0 = mp.problem_from_scene(sc)
    ref, x, res, f = _scipy_vs_restatement(prob, x0, 15, 4)
    assert (res.nfev, res.njev, res.status) == (ref.nfev, ref.njev, ref.status)
    np.testing.assert_allclose(res.cost, ref.cost, rtol=1e-9)
    np.testing.assert_allclose(x, ref.x, rtol=0, atol=1e-7)
    assert np.all(x[2 * prob.C:3 * prob.C] > 0) and np.all(x[2 * prob.C:3 * prob.C] < 1)
    # an x0 outside the bounds is an error, like scipy's ValueError
    bad = x0.copy()
    bad[2 * prob.C] = 1.5
    h = HostHandle(prob)
    with pytest.raises(ValueError):
        h.solve(bad, _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_ANALYTIC, 3))


@pytest.mark.parametrize('name', CASES)
def test_trf_restatement_default_lsmr_first_step(name):
    """scipy's default LSMR settings (atol=btol=1e-6, ~100 iterations): first trial step only."""
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    ref, x, res, f = _scipy_vs_restatement(prob, g['x0'], 2, 0)
    assert (res.nfev, res.njev, res.status) == (ref.nfev, ref.njev, ref.status)
    np.testing.assert_allclose(res.cost, ref.cost, rtol=5e-4)


@pytest.mark.parametrize('name', CASES)
def test_trf_lsmr_vs_reference_result(name):
    """Against the reference's own 10-evaluation result, at the reference's finite-difference noise floor."""
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    oprob, _ = orc.problem_from_scene(scene)
    h = HostHandle(prob)
    opts = _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN, 10)
    x, res, f = h.solve(g['x0'], opts)
    assert res.nfev == int(g['ba10_nfev'])
    rmse = orc.reprojection_rmse(oprob, x)
    # two-sided at the reference's own reproducibility (tests/test_oracle_golden.py), and never worse
    # than the reference beyond that noise
    assert rmse < float(g['ba10_rmse']) + 2.5e-2
    assert res.cost < float(g['ba10_cost']) * (1 + 5e-3)
    # The analytic Jacobian is not the reference's lumped finite-difference Jacobian, so the unconverged
    # 10-evaluation iterate differs (per-detection errors move by a few px at the poorly constrained spline
    # ends); the masks then agree except for detections whose error is near the threshold.
    keep = np.concatenate(orc.outlier_keep_mask(oprob, x, float(g['thres_outlier'])))
    agree = np.mean(keep.astype(np.uint8) == g['outlier_keep'])
    assert agree > 0.95


def test_lm_trust_region_bounds_the_step_host():
    """mvus_solve_opts.lm_trust_radius (the host build of the LM driver, dense normal equations): one trial with a radius far below the
    damped step's length moves x by exactly that radius (the step is cut back along its direction)."""
    from hostcheck_util import HostHandle
    from mvus_amd import problem as mp
    scene, g = load_case('rs_F_2int_3cam')
    prob, x0 = mp.problem_from_scene(scene)

    def run(radius, nfev):
        o = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, nfev)
        o.lm_trust_radius = radius
        x, res, _ = HostHandle(prob).solve(g['x0'], o)      # (a fresh handle: the damping a handle carries from solve to solve starts equal)
        return x, res

    x_free, r_free = run(-1.0, 8)
    assert r_free.cost < r_free.initial_cost
    radius = 1e-4 * float(np.linalg.norm(g['x0']))            # far below the first damped step's length
    x_cut, r_cut = run(radius, 2)
    np.testing.assert_allclose(np.linalg.norm(x_cut - g['x0']), radius, rtol=1e-9)
    assert r_cut.cost < r_cut.initial_cost
    # scipy's rule shrinks the radius to a quarter of the step after any step with actual / predicted < 0.25, whatever it was before:
    # a radius no step reaches at first and scipy's own Delta_0 = |x0| therefore give the same iterates (not those of "no trust region")
    x_big, r_big = run(1e6 * float(np.linalg.norm(g['x0'])), 8)
    x_sci, r_sci = run(0.0, 8)
    assert np.array_equal(x_sci, x_big) and r_sci.nfev == r_big.nfev
    assert r_sci.cost < r_sci.initial_cost
