"""GPU parity tests: the HIP path (through the C ABI, mvus_amd.ba.BAHandle) against the oracle, the
golden vectors captured from the reference, and the host build of the same device math."""
import numpy as np
import pytest

from oracle import ba_oracle as orc
import gauge
from golden_util import CALIB_WP, CASES, CONFIG1_SHAPE, CONVERGED_CASES, ground_truth_x, load_case, reference_spread
from test_fd_mode_host import CONVERGED_RMSE_ATOL, filtered_case, golden_matrix, tie_order_is_the_recorded_one
from mvus_amd import _lib
from mvus_amd import problem as mp

pytestmark = pytest.mark.gpu

RESIDUAL_ATOL = 1e-9        # px, fp64 residuals vs the reference's values
# The converged second BA in the reference's own algorithm is held to SPREAD_FACTOR x what the REFERENCE reproduces of itself
# when its residuals are perturbed in the last place (tests/golden/ens_*.npz: eight runs each with 1e-15 relative noise and
# with noise of one ulp of the pixel coordinates; the larger of the two spreads per quantity).  Measured on MI355X, final
# RMSE minus the reference's / spread: c1 -1.4e-4 / 7.0e-4, rs_F -8.1e-4 / 3.5e-4 (2.3x), calib_KE_bounds -3.1e-3 / 4.3e-2,
# dist -1.4e-4 / 1.1e-4 (1.3x), calib_KE_wellposed -2.4e-5 / 6.2e-4; gauge-invariant distances of x: <= 2.8x (rs_F), else <= 1x.
# (24 ensemble members per noise model on the four small scenes, 4 on the well-posed calibration scene: ~20 min per run there.)
SPREAD_FACTOR = 3.0
JAC_RTOL = 1e-10            # GPU vs host build of the same analytic formulas (relative to column scale)


@pytest.fixture(scope='module')
def BAHandle():
    from mvus_amd.ba import BAHandle as H
    return H


def _host(prob):
    from hostcheck_util import HostHandle
    return HostHandle(prob)


def slots_to_dense(prob, J, ctrl, mJ=None, mctrl=None):
    """Scatter the slot Jacobian into a dense (m, n) matrix in the reference row order."""
    C, P = prob.C, prob.P
    n = prob.n_params
    T = 0 if mJ is None else mJ.shape[1]
    D = np.zeros((2 * prob.M + T, n))
    coff = prob.ctrl_offsets
    for c in range(C):
        a, b = int(prob.det_offsets[c]), int(prob.det_offsets[c + 1])
        Mc = b - a
        cam_cols = [c, C + c, 2 * C + c] + list(range(3 * C + c * P, 3 * C + (c + 1) * P))
        for i in range(a, b):
            g = int(ctrl[i])
            if g < 0:
                continue
            s = int(np.searchsorted(coff, g, side='right') - 1)
            ns, j = int(prob.n_coef[s]), g - int(coff[s])
            cols = cam_cols + [int(prob.spline_x_offsets[s]) + d * ns + j + q for q in range(4) for d in range(3)]
            D[2 * a + (i - a), cols] = J[0, :, i]
            D[2 * a + Mc + (i - a), cols] = J[1, :, i]
    for jrow in range(T):
        for k in range(3):
            g = int(mctrl[k, jrow])
            if g < 0:
                continue
            s = int(np.searchsorted(coff, g, side='right') - 1)
            ns, j = int(prob.n_coef[s]), g - int(coff[s])
            for q in range(4):
                for d in range(3):
                    D[2 * prob.M + jrow, int(prob.spline_x_offsets[s]) + d * ns + j + q] += mJ[12 * k + 3 * q + d, jrow]
    return D


@pytest.mark.parametrize('name', CONVERGED_CASES)
def test_residual_vs_reference_golden(BAHandle, name):
    scene, g = load_case(name)
    prob, x0 = mp.problem_from_scene(scene)
    np.testing.assert_allclose(x0, g['x0'], rtol=0, atol=1e-12)
    oprob, _ = orc.problem_from_scene(scene)
    with BAHandle(prob) as h:
        assert h.n == g['x0'].size and h.m == g['f_x0'].size
        for x, fref in ((g['x0'], g['f_x0']), (g['x0'] + g['delta'], g['f_x0_delta']), (g['ba10_x'], g['ba10_fun'])):
            f = h.residual(x)
            scale = np.maximum(1.0, np.abs(fref))
            assert np.max(np.abs(f - fref) / scale) < RESIDUAL_ATOL
            assert np.array_equal(f == 0, fref == 0)          # visibility: integer information, exact
            fo = orc.residual(oprob, x)
            assert np.max(np.abs(f - fo) / scale) < RESIDUAL_ATOL


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('mode', [_lib.JAC_ANALYTIC, _lib.JAC_PATTERN])
def test_jacobian_operator_vs_host_build(BAHandle, name, mode):
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    hh = _host(prob)
    x = g['x0'] + g['delta']
    pat_host = hh.set_pattern(g['x0'])
    f_ref, D_ref = hh.dense_jacobian(x, mode)
    with BAHandle(prob) as h:
        pat = h.set_pattern(g['x0'])
        assert np.array_equal(pat, pat_host)
        f, J, ctrl = h.residual_jacobian(x, mode)
        mf, mJ, mctrl = h.motion_rows(x, mode)
        D = slots_to_dense(prob, J, ctrl, mJ if h.T else None, mctrl if h.T else None)
        scale = np.maximum(np.abs(D_ref).max(axis=0), 1e-12)
        assert np.max(np.abs(D - D_ref) / scale) < JAC_RTOL
        np.testing.assert_allclose(f, f_ref, rtol=0, atol=1e-9)
        rng = np.random.default_rng(0)
        v, u = rng.normal(size=h.n), rng.normal(size=h.m)
        yref, zref = D_ref @ v, D_ref.T @ u
        np.testing.assert_allclose(h.jv(v), yref, rtol=1e-10, atol=1e-10 * np.abs(yref).max())
        np.testing.assert_allclose(h.jtu(u), zref, rtol=1e-10, atol=1e-10 * np.abs(zref).max())


@pytest.mark.parametrize('name', CONVERGED_CASES)
def test_pattern_vs_reference_matrix(BAHandle, name):
    """jac_BA + compute_visibility (common.py:427-438,490-610), integer: the codes k_pattern computes, against the matrix
    the reference built (golden), row by row.  Bit-exact in every row that is not a twin tie; the flagged rows differ
    from the reference's by exactly the twin control point (same centre knot, the one np.argsort happened to return);
    with the twins decided by np.argsort the whole matrix -- motion rows included -- is the reference's."""
    from mvus_amd import pattern
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    ref = golden_matrix(g)
    with BAHandle(prob) as h:
        pat_c = h.set_pattern(g['x0'])
        mp_c = h.motion_pattern() if h.T else None
        hh = _host(prob)
        assert np.array_equal(pat_c, hh.set_pattern(g['x0']))                  # device == host build of the same rule
        if h.T:
            assert np.array_equal(mp_c, hh.motion_pattern())
        pg, mg = pattern.codes_from_matrix(prob, ref)
        pc, mc = pattern.resolve_ties(prob, g['x0'], pat_c, mp_c, how='canonical')
        differ = pc != pg
        assert np.array_equal(pc < 0, pg < 0)                                   # visibility at x0: exact
        assert pattern.is_tie(pat_c)[differ].all()                              # deviations only in flagged twin rows ...
        assert differ.sum() <= pattern.is_tie(pat_c).sum() <= 0.06 * prob.M
        coff = prob.ctrl_offsets
        for a, b in zip(pc[differ], pg[differ]):                                # ... and only by the twin
            pa = {int(pattern.code_index(a)) + k for k in range(4) if (int(pattern.code_mask(a)) >> k) & 1}
            pb = {int(pattern.code_index(b)) + k for k in range(4) if (int(pattern.code_mask(b)) >> k) & 1}
            (ta,), (tb,) = pa - pb, pb - pa
            s_ = int(np.searchsorted(coff, ta, side='right') - 1)
            t = prob.knots[int(prob.knot_offsets[s_]):int(prob.knot_offsets[s_ + 1])][2:-2]
            assert t[ta - int(coff[s_])] == t[tb - int(coff[s_])]
        # the reference's matrix uploaded as the pattern in force, read back through the masked Jacobian: structure exact
        h.upload_pattern(pg, mg if h.T else None)
        x = g['x0'] + g['delta']
        f, J, ctrl = h.residual_jacobian(x, _lib.JAC_PATTERN)
        mf, mJ, mctrl = h.motion_rows(x, _lib.JAC_PATTERN)
        D = slots_to_dense(prob, J, ctrl, mJ if h.T else None, mctrl if h.T else None)
        assert not D[ref.toarray() == 0].any()                                  # nothing outside the reference's matrix
        if tie_order_is_the_recorded_one():
            pn, mn = pattern.resolve_ties(prob, g['x0'], pat_c, mp_c, how='numpy')
            assert (pattern.reference_pattern(prob, pn, mn if h.T else None) != ref).nnz == 0


@pytest.mark.parametrize('name', CASES)
def test_analytic_jacobian_vs_oracle_central_differences(BAHandle, name):
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    oprob, _ = orc.problem_from_scene(scene)
    oprob.motion_reg = False
    x = g['x0'] + g['delta']
    with BAHandle(prob) as h:
        f, J, ctrl = h.residual_jacobian(x, _lib.JAC_ANALYTIC)
    D = slots_to_dense(prob, J, ctrl)
    Jfd = orc.numeric_jacobian(oprob, x, rel=1e-6)
    if not prob.rs_free:
        Jfd[:, 2 * prob.C:3 * prob.C] = 0.0
    fo = orc.residual(oprob, x)
    ok = np.abs(fo) > 0.05
    alpha, beta, rs, cams, tck = orc.unpack_x(oprob, x)
    near = []
    for c in range(prob.C):
        tau = orc.detection_to_global(oprob, c, alpha, beta, rs, cams[c])[0]
        ne = (np.abs(tau[:, None] - oprob.interval.reshape(1, -1)) < 0.05).any(axis=1)
        near += [ne, ne]
    ok &= ~np.concatenate(near)
    scale = np.maximum(np.abs(Jfd[ok]).max(axis=0), 1e-12)
    assert np.max(np.abs(D[ok] - Jfd[ok]) / scale) < 2e-5


@pytest.mark.parametrize('name', CONVERGED_CASES)
def test_outlier_mask_bit_exact(BAHandle, name):
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    with BAHandle(prob) as h:
        keep = h.outlier_mask(g['ba10_x'], float(g['thres_outlier']))
    assert np.array_equal(keep.astype(np.uint8), g['outlier_keep'])


@pytest.mark.parametrize('name', CASES)
def test_trf_lsmr_gpu_matches_host_backend(BAHandle, name):
    """Same templated optimiser, HIP backend vs plain-C++ backend, LSMR capped where rounding cannot fork it."""
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    opts = _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN, 15)
    opts.lsmr_maxiter = 4
    xh, rh, fh = _host(prob).solve(g['x0'], opts)
    with BAHandle(prob) as h:
        r = h.solve(g['x0'], opts=opts)
    assert (r.nfev, r.njev, r.status) == (rh.nfev, rh.njev, rh.status)
    np.testing.assert_allclose(r.cost, rh.cost, rtol=1e-9)
    np.testing.assert_allclose(r.x, xh, rtol=0, atol=1e-7)
    np.testing.assert_allclose(r.fun, fh, rtol=0, atol=1e-6)


@pytest.mark.parametrize('name', CASES)
def test_ba_vs_reference_result(BAHandle, name):
    """Scene.BA's 10-evaluation result: never worse than the reference beyond its own reproducibility."""
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    oprob, _ = orc.problem_from_scene(scene)
    with BAHandle(prob) as h:
        r = h.solve(g['x0'], solver=_lib.SOLVER_TRF_LSMR, jac_mode=_lib.JAC_PATTERN, max_nfev=10, matrix=golden_matrix(g))
        keep = h.outlier_mask(r.x, float(g['thres_outlier']))
    assert r.nfev == int(g['ba10_nfev'])
    # analytic derivatives in the reference's matrix instead of its lumped forward differences: another (better) iterate
    # after 10 evaluations -- one-sided bound; the two-sided comparison is test_fd_mode_ba_vs_reference_result
    assert r.cost < float(g['ba10_cost']) * (1 + 5e-3)
    assert orc.reprojection_rmse(oprob, r.x) < float(g['ba10_rmse']) + 2.5e-2
    assert np.mean(keep.astype(np.uint8) == g['outlier_keep']) > 0.95


def test_error_paths(BAHandle):
    scene, g = load_case('calib_KE_bounds_3cam')
    prob, x0 = mp.problem_from_scene(scene)
    with BAHandle(prob) as h:
        bad = x0.copy()
        bad[2 * prob.C] = 1.5                         # outside 0 <= rs <= 1: scipy raises ValueError
        with pytest.raises(ValueError):
            h.solve(bad)
        nanx = x0.copy()
        nanx[0] = np.nan
        with pytest.raises(ValueError):
            h.solve(nanx)
        with pytest.raises(ValueError):
            h.residual(x0[:-1])
    broken = mp.problem_from_scene(scene)[0]
    broken.interval = broken.interval[::-1].copy()    # end < start
    with pytest.raises(ValueError):
        BAHandle(broken)


def test_empty_and_ragged_cameras(BAHandle):
    """A camera without detections and a camera whose detections all fall outside the spline."""
    from mvus_amd import synth
    sc = synth.make_scene(3, 900, seed=9, rolling_shutter=True, knot_spacing=14.0)
    sc.detections[1] = sc.detections[1][:, :0]
    sc.detections[2] = sc.detections[2].copy()
    sc.detections[2][0] += 1e6
    prob, x0 = mp.problem_from_scene(sc)
    oprob, _ = orc.problem_from_scene(sc)
    with BAHandle(prob) as h:
        f = h.residual(x0)
        fo = orc.residual(oprob, x0)
        np.testing.assert_allclose(f, fo, rtol=0, atol=1e-9)
        assert not f[2 * prob.det_offsets[2]:].any()
        r = h.solve(x0, jac_mode=_lib.JAC_ANALYTIC, max_nfev=5)
        assert np.isfinite(r.cost) and r.cost <= r.initial_cost
        # the LM / Schur path: the empty camera's block is all zero (damped by the unit diagonal), its parameters stay put
        r2 = h.solve(x0, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=8)
        assert np.isfinite(r2.cost) and r2.cost < r2.initial_cost
        C, P = prob.C, prob.P
        for cam in (1, 2):
            cols = [cam, C + cam, 2 * C + cam] + list(range(3 * C + cam * P, 3 * C + (cam + 1) * P))
            np.testing.assert_array_equal(r2.x[cols], x0[cols])


def test_full_size_properties(BAHandle):
    """BASELINE config 3 (32 cams x 500k obs): size-independent properties of the operator pair."""
    from mvus_amd import synth
    sc = synth.baseline_scene(2)
    prob, x0 = mp.problem_from_scene(sc)
    assert prob.C == 32 and abs(prob.M - 500_000) < 25_000
    rng = np.random.default_rng(0)
    oprob, ox0 = orc.problem_from_scene(sc)
    np.testing.assert_array_equal(x0, ox0)
    with BAHandle(prob) as h:
        f1 = h.residual(x0)
        # the headline workload against the oracle, all 32 cameras x 504k detections: residual (zero pattern exact) and the integer
        # outlier mask (bit-exact), at x0 and at the point a short LM solve reaches, whose cost the oracle re-evaluates
        fo = orc.residual(oprob, x0)
        assert f1.shape == fo.shape and np.array_equal(f1 == 0, fo == 0)
        assert np.max(np.abs(f1 - fo) / np.maximum(1.0, np.abs(fo))) < 1e-9
        assert np.array_equal(h.outlier_mask(x0, 10.0), np.concatenate(orc.outlier_keep_mask(oprob, x0, 10.0)))
        r = h.solve(x0, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=6)
        fo2 = orc.residual(oprob, r.x)
        assert r.cost < r.initial_cost and abs(0.5 * float(fo2 @ fo2) - r.cost) <= 1e-9 * r.cost
        assert np.max(np.abs(r.fun - fo2) / np.maximum(1.0, np.abs(fo2))) < 1e-9
        assert np.array_equal(h.outlier_mask(r.x, 10.0), np.concatenate(orc.outlier_keep_mask(oprob, r.x, 10.0)))
        del fo, fo2, r
        f2, J, ctrl = h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        assert np.array_equal(f1, f2)                              # idempotent, and both kernels agree bit for bit
        assert (ctrl >= 0).mean() > 0.9
        del J
        v1, v2, u = rng.normal(size=h.n), rng.normal(size=h.n), rng.normal(size=h.m)
        y1, y2, y12 = h.jv(v1), h.jv(v2), h.jv(v1 + 2.0 * v2)
        np.testing.assert_allclose(y12, y1 + 2.0 * y2, rtol=1e-9, atol=1e-9 * np.abs(y12).max())   # linearity
        z = h.jtu(u)
        lhs, rhs = float(y1 @ u), float(v1 @ z)
        assert abs(lhs - rhs) <= 1e-9 * np.linalg.norm(y1) * np.linalg.norm(u)                       # adjoint pair
        # directional derivative: |r| is smooth away from r = 0
        d = rng.normal(size=h.n) * 1e-3
        eps = 1e-4
        fp, fm = h.residual(x0 + eps * d), h.residual(x0 - eps * d)
        jd = h.jv(d)
        ok = (np.abs(f1) > 0.5) & (fp != 0) & (fm != 0)
        err = np.abs((fp - fm)[ok] / (2 * eps) - jd[ok])
        assert np.quantile(err / (np.abs(jd[ok]) + 1e-3), 0.999) < 1e-3
        # outlier mask == threshold on the residual pairs, bit for bit
        # normal equations of the LM path: gradient = J^T f (assembled from the same slot Jacobian by other kernels)
        h.residual_jacobian(x0, _lib.JAC_ANALYTIC)                 # (the residual calls above overwrote the held f)
        gg, A, band, cross = h.normal_equations()
        z_f = h.jtu(f1)
        np.testing.assert_allclose(gg, z_f, rtol=0, atol=1e-9 * np.abs(z_f).max())
        # ... and the damped step of the whole GPU solve chain at this size (155 separators, 8 levels of cyclic reduction)
        # against LAPACK: banded Cholesky of the spline block, dense Schur complement
        from lm_reference import lapack_lm_step
        lam = 0.5
        p_gpu = h.lm_step(lam)
        p_ref = lapack_lm_step(prob, gg, A, band, cross, lam)
        np.testing.assert_allclose(p_gpu, p_ref, rtol=0, atol=1e-6 * np.abs(p_ref).max())
        del A, band, cross
        keep = h.outlier_mask(x0, 10.0)
        off = prob.det_offsets
        ex = np.concatenate([f1[2 * a:2 * a + (b - a)] for a, b in zip(off[:-1], off[1:])])
        ey = np.concatenate([f1[2 * a + (b - a):2 * b] for a, b in zip(off[:-1], off[1:])])
        assert np.array_equal(keep, np.sqrt(ex ** 2 + ey ** 2) < 10.0)


@pytest.mark.parametrize('name', CASES)
def test_fd_jacobian_gpu_vs_host_and_scipy(BAHandle, name):
    """MVUS_JAC_FD: grouped forward differences on the GPU = the host build = scipy's approx_derivative."""
    from mvus_amd import pattern
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    hh = _host(prob)
    pat, groups, ng = hh.prepare_fd(g['x0'])
    x = g['x0'] + g['delta']
    if prob.rs_bounds:
        x[2 * prob.C:3 * prob.C] = np.clip(x[2 * prob.C:3 * prob.C], 0.0, 1.0)
    f_ref, D_ref = hh.dense_jacobian(x, _lib.JAC_FD)
    with BAHandle(prob) as h:
        assert h.prepare_fd(g['x0']) == ng
        f, J, ctrl = h.residual_jacobian(x, _lib.JAC_FD)
        mf, mJ, mctrl = h.motion_rows(x, _lib.JAC_FD)
    D = slots_to_dense(prob, J, ctrl, mJ if prob.motion_reg else None, mctrl if prob.motion_reg else None)
    scale = np.maximum(np.abs(D_ref).max(axis=0), 1e-12)
    # both divide differences of residuals that agree to ~1e-13 px by h ~ 1.5e-8: the quotient agrees to ~1e-5
    # (per column the noise floor is ~1e-13 px / 1.5e-8 = 1e-5 absolute; weak columns such as k3 have small scale)
    assert np.max(np.abs(D - D_ref) / np.maximum(scale, 1e-2 * np.abs(D_ref).max())) < 1e-3
    assert np.quantile(np.abs(D - D_ref) / scale, 0.999) < 1e-2
    A = pattern.reference_pattern(prob, pat, hh.motion_pattern() if prob.motion_reg else None)
    assert not D[A.toarray() == 0].any()                      # nothing outside the reference pattern


@pytest.mark.parametrize('passes', ['two', 'one'])
@pytest.mark.parametrize('name', CASES)
def test_fd_mode_ba_vs_reference_result(BAHandle, name, passes, monkeypatch):
    """The reference's own algorithm end to end on the GPU (scipy TRF + LSMR + grouped 2-point differences), at the point every real
    call returns: max_iter = 10 (Scene.BA's default, common.py:441; main.py:49).  The unconverged iterate is chaotic in the last bits
    of LSMR's sums, for the reference as for anybody else -- so the bars are THE REFERENCE'S OWN: its 10-evaluation BA re-run with one
    unit in the last place of noise on the residuals (tests/golden/ens10_<case>.npz, 16 members) moves by `spread` in cost, RMSE and
    inlier-mask flips, and the GPU is held to SPREAD_FACTOR x that (round 6; rounds 1-5 used multiples of the GPU's own deviations).
    passes = 'one': LSMR with ONE pass over J per iteration (MVUS_LSMR_ONE_PASS=1) -- inside the bars on three scenes, outside on
    dist_fixed_2cam (cost 3.1e-3 against 3 x 4.8e-4, 14 flips against 3 x 1): that is why it is opt-in, and this test keeps the verdict."""
    from golden_util import reference_spread_10
    if passes == 'one':
        monkeypatch.setenv('MVUS_LSMR_ONE_PASS', '1')
    else:
        monkeypatch.delenv('MVUS_LSMR_ONE_PASS', raising=False)
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    oprob, _ = orc.problem_from_scene(scene)
    with BAHandle(prob) as h:
        r = h.solve(g['x0'], solver=_lib.SOLVER_TRF_LSMR, jac_mode=_lib.JAC_FD, max_nfev=10, matrix=golden_matrix(g))
        keep = h.outlier_mask(r.x, float(g['thres_outlier']))
    assert r.nfev == int(g['ba10_nfev'])
    sp = reference_spread_10(name)['ba10']
    d_cost = abs(r.cost - float(g['ba10_cost'])) / float(g['ba10_cost'])
    d_rmse = abs(orc.reprojection_rmse(oprob, r.x) - float(g['ba10_rmse']))
    flips = int(np.sum(keep.astype(np.uint8) != g['outlier_keep']))
    print('FD 10 evaluations (%s-pass LSMR) %s: cost rel %.2e (reference spread %.2e), rmse %.2e px (%.2e), %d mask flips (%d)'
          % (passes, name, d_cost, sp['cost'], d_rmse, sp['rmse'], flips, sp['flips']))
    inside = d_cost <= SPREAD_FACTOR * sp['cost'] and d_rmse <= SPREAD_FACTOR * sp['rmse'] and flips <= SPREAD_FACTOR * sp['flips']
    if passes == 'one' and name == 'dist_fixed_2cam':
        # the one place the one-pass arithmetic leaves the reference's own bars: recorded, not hidden (and the reason it is opt-in);
        # should a change of the kernels bring it inside, this assertion says so and the default can be reconsidered
        assert not inside, 'one-pass LSMR is now inside the reference-derived bars on every fixture: reconsider the default'
        return
    assert d_cost <= SPREAD_FACTOR * sp['cost']
    assert d_rmse <= SPREAD_FACTOR * sp['rmse']
    assert flips <= SPREAD_FACTOR * sp['flips'], flips


@pytest.mark.parametrize('passes', ['two', 'one'])
@pytest.mark.parametrize('name', CONVERGED_CASES)
def test_converged_second_ba_fd_mode(BAHandle, name, passes, monkeypatch):
    """North-star parity at the one point where it is decidable: the reference's second BA (main.py:59) run with max_iter=200
    (status 3 on the four small scenes; the well-posed calibration scene uses all 200 evaluations, status 0).  The reference's
    algorithm on the GPU -- TRF + LSMR + grouped forward differences over the reference's matrix -- from the reference's
    start.  Compared with the reference's result: same status, the ANSWER ITSELF (res.x is all the caller reads,
    common.py:672-695) in gauge-invariant terms -- trajectory at the detection time stamps and camera centres / orientations
    after the best similarity, beta differences, alpha ratios, rs, and K, d with opt_calib (tests/gauge.py) -- and the final
    RMSE, each within SPREAD_FACTOR x the reference's own reproducibility; inlier mask at that point identical.
    passes = 'one': the LSMR iteration with ONE pass over J (MVUS_LSMR_ONE_PASS=1, round 5) -- another order of the same sums; the
    converged answer must sit inside the same bars."""
    if passes == 'one':
        monkeypatch.setenv('MVUS_LSMR_ONE_PASS', '1')
    else:
        monkeypatch.delenv('MVUS_LSMR_ONE_PASS', raising=False)
    scene, g = filtered_case(name)
    prob, _ = mp.problem_from_scene(scene)
    oprob, _ = orc.problem_from_scene(scene)
    with BAHandle(prob) as h:
        r = h.solve(g['ba2_200_x0'], solver=_lib.SOLVER_TRF_LSMR, jac_mode=_lib.JAC_FD, max_nfev=200, matrix=golden_matrix(g, second=True))
        keep = h.outlier_mask(r.x, float(g['thres_outlier']))
    spread = reference_spread(oprob, name, g['ba2_200_x'])
    c = gauge.compare(oprob, g['ba2_200_x'], r.x)
    d_rmse = c['rmse_b'] - float(g['ba2_200_rmse'])
    print('converged FD (%s-pass LSMR) %s: rmse %+.2e px (spread %.1e), cost %.9g vs %.9g, nfev %d vs %d, status %d; x vs reference / spread: %s'
          % (passes, name, d_rmse, spread['rmse'], r.cost, float(g['ba2_200_cost']), r.nfev, int(g['ba2_200_nfev']), r.status,
             ' '.join('%s %.1e/%.1e' % (k, c[k], spread[k]) for k in spread if k != 'rmse')))
    assert r.status == int(g['ba2_200_status'])
    assert abs(r.nfev - int(g['ba2_200_nfev'])) <= 6
    assert abs(c['rmse_a'] - float(g['ba2_200_rmse'])) < 1e-9                    # the oracle reproduces the reference's own figure
    assert abs(d_rmse) <= SPREAD_FACTOR * spread['rmse']
    for k in spread:
        if k != 'rmse':
            assert c[k] <= SPREAD_FACTOR * spread[k] + 1e-12, (k, c[k], spread[k])
    assert np.array_equal(keep.astype(np.uint8), g['ba2_200_keep'])


# LM + Schur (the fast solver, settings['ba_solver'] = 'lm') is ANOTHER estimator of the same scene: it minimises the same
# objective with the exact Jacobian and reaches a lower value than the reference's stopping point, so its x is not the
# reference's x.  What can be stated -- and is, here -- is how far apart the two are next to how far each is from the truth the
# synthetic scene was generated from (two estimates of one truth differ by about their errors).  Bounds = measured on MI355X x 1.5
# (trajectory RMS in metres after similarity alignment, at the detection time stamps; scene extent ~20 m; 200 evaluations,
# damping floor 3e-3):
#                         LM vs reference   reference vs truth   LM vs truth
#   c1_pinhole_2cam            0.33              0.23               0.20        (without the damping floor: 3.9 / 3.8)
#   rs_F_2int_3cam             0.032             0.058              0.079
#   dist_fixed_2cam            0.10              0.12               0.095
#   calib_KE_wellposed_5cam    0.0080            0.0059             0.0078      K within 9.1e-3 relative, d within 0.059 of the reference's
#   config1_shape_7cam         0.033             0.057              0.059
# calib_KE_bounds_3cam is ill posed (the distortion coefficients are unobservable: the reference's own first BA takes k1 from
# -0.03 to 27.7 and its converged K, d do not reproduce): no exact-Jacobian solver has a meaningful answer there, LM's k3 reaches
# 5e4.  Scene.BA therefore never picks LM by itself (tests/test_host_logic.py::test_scene_default_is_the_reference_algorithm).
LM_TRAJ_RMS_VS_REF = {'c1_pinhole_2cam': 0.50, 'rs_F_2int_3cam': 0.048, 'dist_fixed_2cam': 0.15, CALIB_WP: 0.012, CONFIG1_SHAPE: 0.050}
LM_TRAJ_RMS_VS_TRUTH_FACTOR = 1.5          # LM's distance to the truth <= this x the reference's distance to the truth


@pytest.mark.parametrize('name', CONVERGED_CASES)
@pytest.mark.parametrize('mode', ['trf_pattern', 'lm_schur'])
def test_converged_second_ba_analytic_modes(BAHandle, name, mode):
    """The analytic-Jacobian solvers from the same start: they do not stop where the reference's trust region collapses
    (its Jacobian lumps the dropped control point into the kept ones) but go on to a LOWER value of the same objective.
    Asserted: the oracle confirms the objective value; the cost is not above the reference's; the inlier mask at the end;
    and for LM the recovered solution itself against the reference's AND against ground truth (table above), the damping
    floor keeping it out of the directions the data does not determine (without it: 3.9 m on c1)."""
    scene, g = filtered_case(name)
    prob, _ = mp.problem_from_scene(scene)
    oprob, _ = orc.problem_from_scene(scene)
    ref_cost, ref_rmse = float(g['ba2_200_cost']), float(g['ba2_200_rmse'])
    with BAHandle(prob) as h:
        if mode == 'trf_pattern':
            r = h.solve(g['ba2_200_x0'], solver=_lib.SOLVER_TRF_LSMR, jac_mode=_lib.JAC_PATTERN, max_nfev=200, matrix=golden_matrix(g, second=True))
        else:
            r = h.solve(g['ba2_200_x0'], solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=200)
        keep = h.outlier_mask(r.x, float(g['thres_outlier']))
    fo = orc.residual(oprob, r.x)
    assert abs(0.5 * float(fo @ fo) - r.cost) < 1e-9 * r.cost
    rmse = orc.reprojection_rmse(oprob, r.x)
    flips = int(np.sum(keep.astype(np.uint8) != g['ba2_200_keep']))
    c = gauge.compare(oprob, g['ba2_200_x'], r.x)
    xt = ground_truth_x(oprob, name)
    ct_ref, ct = gauge.compare(oprob, xt, g['ba2_200_x']), gauge.compare(oprob, xt, r.x)
    print('converged %s %s: cost %.9g vs ref %.9g (%.2f %%), rmse %.6f vs %.6f, nfev %d status %d, %d mask flips; trajectory rms: vs ref %.2e, '
          'ref vs truth %.2e, this vs truth %.2e; centres vs truth %.2e (ref %.2e)'
          % (mode, name, r.cost, ref_cost, 100 * (r.cost / ref_cost - 1), rmse, ref_rmse, r.nfev, r.status, flips, c['traj_rms'],
             ct_ref['traj_rms'], ct['traj_rms'], ct['centre_max'], ct_ref['centre_max']))
    assert r.cost <= ref_cost * (1 + 1e-6)
    # measured (cost vs reference): -2.1/-2.5 % c1, -14/-7.5 % rs_F, -43/-76 % calib_KE_bounds, -0.9/-0.8 % dist, -0.1/-1.1 % calib_KE_wellposed
    if not prob.motion_reg:
        assert rmse <= ref_rmse + 1e-4           # without a regulariser in the objective lower cost IS lower reprojection error
    assert flips <= 3
    if mode == 'lm_schur' and name in LM_TRAJ_RMS_VS_REF:
        assert c['traj_rms'] <= LM_TRAJ_RMS_VS_REF[name]
        assert ct['traj_rms'] <= LM_TRAJ_RMS_VS_TRUTH_FACTOR * ct_ref['traj_rms']
        assert np.all(np.isfinite(r.x))
        if prob.opt_calib:                       # K, d stay physical and near the reference's (well-posed scene)
            assert c['K_rel_max'] <= 2e-2 and c['d_max'] <= 0.1, (c['K_rel_max'], c['d_max'])
            assert ct['K_rel_max'] <= 2e-2 and ct['d_max'] <= 0.15, (ct['K_rel_max'], ct['d_max'])


@pytest.mark.parametrize('name', CASES)
def test_remove_outliers_in_place_equals_fresh_handle(BAHandle, name):
    """mvus_ba_remove_outliers: mask bit-exact vs the reference, and the compacted resident problem behaves exactly
    like a new handle built from the filtered detections (residual bit for bit, same BA result)."""
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    x = g['ba10_x']
    thres = float(g['thres_outlier'])
    with BAHandle(prob) as h:
        keep = h.remove_outliers(x, thres)
        assert np.array_equal(keep.astype(np.uint8), g['outlier_keep'])
        off = g['det_offsets']
        for i in range(scene.num_cam):
            scene.detections[i] = scene.detections[i][:, keep[off[i]:off[i + 1]]]
        prob2, _ = mp.problem_from_scene(scene)
        assert np.array_equal(h.prob.det_offsets, prob2.det_offsets) and np.array_equal(h.prob.frame, prob2.frame)
        with BAHandle(prob2) as h2:
            assert np.array_equal(h.residual(x), h2.residual(x))
            opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 6)
            r1, r2 = h.solve(x, opts=opts), h2.solve(x, opts=opts)
            assert (r1.nfev, r1.njev, r1.status) == (r2.nfev, r2.njev, r2.status)
            np.testing.assert_allclose(r1.cost, r2.cost, rtol=1e-10)
            np.testing.assert_allclose(r1.x, r2.x, rtol=0, atol=1e-8 * max(1.0, np.abs(r2.x).max()))
        # a second pass removes nothing new at the same x and threshold... unless the fit moved: here x is unchanged
        keep2 = h.remove_outliers(x, thres)
        assert keep2.all()


def test_opt_sync_off_freezes_alpha_and_beta(BAHandle):
    """settings['opt_sync'] = False (common.py:512-515) on the GPU: no Jacobian entries for alpha / beta in any mode, and
    both solvers leave them where they were."""
    scene, g = load_case('rs_F_2int_3cam')
    scene.settings['opt_sync'] = False
    prob, x0 = mp.problem_from_scene(scene)
    C = prob.C
    with BAHandle(prob) as h:
        for mode in (_lib.JAC_ANALYTIC, _lib.JAC_PATTERN):
            h.prepare_pattern(x0)
            f, J, ctrl = h.residual_jacobian(x0 + g['delta'], mode)
            assert not J[:, :2, :].any()                                   # slots 0, 1 = alpha, beta
        for solver, jm in ((_lib.SOLVER_TRF_LSMR, _lib.JAC_FD), (_lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN), (_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC)):
            r = h.solve(x0, solver=solver, jac_mode=jm, max_nfev=14)      # (LM rejects its first five trials from this start)
            np.testing.assert_array_equal(r.x[:2 * C], x0[:2 * C])
            assert r.cost < r.initial_cost


@pytest.mark.parametrize('total_obs', [150, 1900, 16400, 131500])
def test_chunk_orders_cover_every_chunk(BAHandle, total_obs):
    """The XCD-aware tile orders pad their grids (to multiples of 8 x run) and skip tiles past the end: with 1, ~8, ~65 and ~515
    chunks every detection must still be evaluated, stored (chunk-major J, exported slot-major), multiplied (J v, J^T u) and
    assembled.  Checked through identities that hold for any x: J v against central differences of the residual along v,
    u.(J v) == (J^T u).v, the normal-equation gradient == J^T f, and all of it against the host build where it is small."""
    from mvus_amd import synth
    sc = synth.make_scene(2, total_obs, seed=total_obs % 97, knot_spacing=15.0)
    prob, x0 = mp.problem_from_scene(sc)
    rng = np.random.default_rng(1)
    with BAHandle(prob) as h:
        f, J, ctrl = h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        assert ctrl.shape == (prob.M,) and J.shape[-1] == prob.M
        visible = ctrl >= 0
        assert visible.sum() > 0.9 * prob.M                      # every chunk evaluated: no block of untouched rows
        v, u = rng.normal(size=h.n) * 1e-3, rng.normal(size=h.m)
        if not prob.rs_free:
            v[2 * prob.C:3 * prob.C] = 0.0                        # a frozen column: J has no entry for it, the residual still depends on rs
        Jv = h.jv(v)
        eps = 1e-4
        fp, fm = np.abs(h.residual(x0 + eps * v)), np.abs(h.residual(x0 - eps * v))
        fd = (fp - fm) / (2 * eps)
        sel = np.concatenate((visible, visible)) if h.T == 0 else np.concatenate((visible, visible, np.ones(h.T, bool)))
        ok = sel & (np.abs(f) > 1e-3 + 4 * eps * np.abs(Jv)) & (fp > 0) & (fm > 0)   # |r| has a kink at 0 (not to be crossed between the two evaluations); a time stamp may leave its interval
        assert ok.sum() > 0.9 * sel.sum()
        assert np.max(np.abs(Jv[ok] - fd[ok])) < 1e-5 * max(1.0, np.abs(Jv).max())
        JTu = h.jtu(u)
        assert abs(u @ Jv - JTu @ v) < 1e-9 * max(1.0, abs(u @ Jv))
        f, J, ctrl = h.residual_jacobian(x0, _lib.JAC_ANALYTIC)    # the residual calls above replaced the f the handle holds
        g = h.normal_equations()[0]                              # assembled from the J held: gradient == J^T f
        np.testing.assert_allclose(g, h.jtu(f), rtol=1e-9, atol=1e-9 * np.abs(g).max())
        if total_obs <= 2000:
            hh = _host(prob)
            hh.set_pattern(x0)
            f_ref, D_ref = hh.dense_jacobian(x0, _lib.JAC_ANALYTIC)
            np.testing.assert_allclose(Jv, D_ref @ v, rtol=1e-10, atol=1e-12)
            np.testing.assert_allclose(JTu, D_ref.T @ u, rtol=1e-10, atol=1e-10 * np.abs(JTu).max())
        r = h.solve(x0, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=4)
        assert r.cost < r.initial_cost


@pytest.mark.parametrize('name,jac', [('calib_KE_bounds_3cam', 'fd'), ('calib_KE_bounds_3cam', 'pattern'), (CALIB_WP, 'pattern')])
def test_bounded_lsmr_on_the_device_is_the_host_driven_loop(BAHandle, name, jac, monkeypatch):
    """Round 6: the LSMR iteration of the BOUNDED problem (rs_bounds: scipy's trf_bounds works on A = [J diag(D); diag(E)], common.py:654-670)
    runs device resident like the unbounded one -- column scaling, the n extra rows of u and the scalar recurrences in kernels, batches
    of eight iterations without a host round trip -- instead of three synchronisations and ~17 launches per iteration.  Same operations
    in the same order: the iterates are the host-driven loop's (MVUS_LSMR_BOUNDED_HOST=1) bit for bit."""
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    assert prob.rs_bounds
    jm = _lib.JAC_FD if jac == 'fd' else _lib.JAC_PATTERN

    def run():
        with BAHandle(prob) as h:
            return h.solve(g['x0'], solver=_lib.SOLVER_TRF_LSMR, jac_mode=jm, max_nfev=8, matrix=golden_matrix(g))

    monkeypatch.delenv('MVUS_LSMR_BOUNDED_HOST', raising=False)
    dev = run()
    monkeypatch.setenv('MVUS_LSMR_BOUNDED_HOST', '1')
    host = run()
    print('%s %s: %d LSMR iterations in %d evaluations, %.1f ms (device) against %.1f ms (host-driven)' % (name, jac, dev.lin_iters, dev.nfev, dev.solve_ms, host.solve_ms))
    assert (dev.nfev, dev.status, dev.lin_iters) == (host.nfev, host.status, host.lin_iters)
    assert dev.cost == host.cost and np.array_equal(dev.x, host.x)
    assert dev.cost < dev.initial_cost
