"""The reduced camera system by blocked L D L^T in matrix-core block images (mvus_amd/csrc/ba_rcs.hip.h; the default since round 5)
against the block Gauss-Jordan of rounds 1 - 4 (MVUS_RCS=gj) and against a dense LAPACK solve of the same damped system; the block
rows below each super-block solved inside the factor launch (flag hand-over between workgroups) and by a launch of their own
(MVUS_RCS_TRSM=launch).

No counterpart in the reference (reconstruction/common.py:670 delegates the whole solve to scipy); what is checked is that the
damped step p of the whole GPU chain solves (H + lambda diag H) p = -g, for reduced systems of 1 .. 5 super-panels of 144 unknowns,
sizes that are not multiples of 16 (identity padding), and P = 6 and 15 camera parameters.  Also here: the band solver's one-rank path
of round 5 (no right-hand-side copy, no back-correction) against the path time shards still run and against the dense solve."""
import numpy as np
import pytest

from mvus_amd import _lib
from mvus_amd import problem as mp

pytestmark = pytest.mark.gpu


def _dense_step(prob, h, lam):
    """(H + lam diag H) p = -g with H, g from mvus_ba_normal_equations, solved densely on the host."""
    from test_gpu_schur import internal_index
    g, A, band, cross = h.normal_equations()
    n, C, B, W = prob.n_params, prob.C, 3 + prob.P, band.shape[1]
    N = int(prob.n_coef.sum())
    cam_idx, spl_idx = internal_index(prob)
    H = np.zeros((n, n))
    for c in range(C):
        H[np.ix_(cam_idx[c], cam_idx[c])] = A[c]
    E = cross.reshape(C * B, 3 * N)
    ci = cam_idx.ravel()
    H[np.ix_(ci, spl_idx)] = E
    H[np.ix_(spl_idx, ci)] = E.T
    for w in range(W):
        for gi in range(N - w):
            ri, cj = spl_idx[3 * gi:3 * gi + 3], spl_idx[3 * (gi + w):3 * (gi + w) + 3]
            H[np.ix_(ri, cj)] = band[gi, w]
            if w > 0:
                H[np.ix_(cj, ri)] = band[gi, w].T
    d = np.diag(H).copy()
    d = np.where(d > 0, d, 1.0)
    return np.linalg.solve(H + lam * np.diag(d), -g)


# cameras x calib: nn = C * 9 or C * 18 -> 16-blocks / super-panels of 9 blocks
@pytest.mark.parametrize('cams,calib', [(2, False), (7, False), (15, False), (16, False), (17, False), (32, False), (33, False),
                                       (48, False), (64, False), (5, True), (9, True), (40, True)])
def test_damped_step_of_the_blocked_ldlt(cams, calib, monkeypatch):
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    sc = synth.make_scene(cams, 150 * cams + 400, seed=100 + cams, rolling_shutter=True, num_knots=24, opt_calib=calib)
    prob, x0 = mp.problem_from_scene(sc)
    nn = prob.C * (3 + prob.P)
    lams = (1e-3, 0.7)
    monkeypatch.delenv('MVUS_RCS', raising=False)
    with BAHandle(prob) as h:
        h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        steps = [h.lm_step(lam) for lam in lams]
        again = h.lm_step(lams[0])
        refs = [_dense_step(prob, h, lam) for lam in lams]
    monkeypatch.setenv('MVUS_RCS_TRSM', 'launch')
    with BAHandle(prob) as h:
        h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        own = [h.lm_step(lam) for lam in lams]
    monkeypatch.delenv('MVUS_RCS_TRSM')
    for p, q in zip(steps, own):
        assert np.array_equal(p, q), 'rows below the super-block: in-launch and separate-launch forms differ'
    monkeypatch.setenv('MVUS_RCS', 'gj')
    with BAHandle(prob) as h:
        h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        gj = [h.lm_step(lam) for lam in lams]
    assert np.array_equal(steps[0], again), 'the solve is not reproducible run to run'
    for lam, p, q, r in zip(lams, steps, gj, refs):
        scale = np.abs(r).max()
        print('nn = %d, lambda = %g: |p - dense| = %.2e, |gj - dense| = %.2e (relative to max |p|)'
              % (nn, lam, np.abs(p - r).max() / scale, np.abs(q - r).max() / scale))
        np.testing.assert_allclose(p, r, rtol=0, atol=1e-7 * scale)
        np.testing.assert_allclose(p, q, rtol=0, atol=1e-7 * scale)


def test_non_positive_pivot_raises_the_flag_and_lm_recovers():
    """A camera without detections has a zero block: the damping's max(diag, 1) fallback keeps the system definite, and a solve with
    lambda = 0 on it must fail through the flag (LM then raises lambda) rather than return garbage."""
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    sc = synth.make_scene(4, 1500, seed=5, rolling_shutter=True, num_knots=16)
    prob, x0 = mp.problem_from_scene(sc)
    with BAHandle(prob) as h:
        r = h.solve(x0, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=8)
        assert np.isfinite(r.cost) and r.cost < r.initial_cost


def test_handover_timeout_repeats_the_solve_on_the_other_route(monkeypatch, capfd):
    """The workgroups of k_rcs_factor that solve the rows below a super-block wait for the factor workgroup's flag with a BOUNDED spin.
    A time-out (a GPU shared with other processes: forward progress between workgroups of one launch is not guaranteed) is not a lost
    pivot: the driver must repeat the SAME solve through the separate-launch route, not raise the damping.  MVUS_RCS_SPIN_LIMIT=0 makes
    the first poll give up; the iterates must be those of an undisturbed handle, bit for bit."""
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    sc = synth.make_scene(17, 17 * 150 + 400, seed=117, rolling_shutter=True, num_knots=24)       # 153 unknowns: one block row below the first super-panel
    prob, x0 = mp.problem_from_scene(sc)
    kw = dict(solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=6)
    for k in ('MVUS_RCS', 'MVUS_RCS_TRSM', 'MVUS_RCS_SPIN_LIMIT', 'MVUS_DEBUG'):
        monkeypatch.delenv(k, raising=False)
    with BAHandle(prob) as h:
        ref = h.solve(x0, **kw)
        h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        p_ref = h.lm_step(1e-3)
    capfd.readouterr()
    monkeypatch.setenv('MVUS_RCS_SPIN_LIMIT', '0')
    monkeypatch.setenv('MVUS_DEBUG', '1')
    with BAHandle(prob) as h:
        r = h.solve(x0, **kw)
    err = capfd.readouterr().err
    with BAHandle(prob) as h:
        h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        p = h.lm_step(1e-3)                                # the inspection hook repeats the solve as well
    err2 = capfd.readouterr().err
    assert err.count('hand-over time-out') == 1 and err2.count('hand-over time-out') == 1, (err, err2)   # once per handle: the route is switched for good
    assert np.array_equal(r.x, ref.x) and (r.nfev, r.status, r.cost) == (ref.nfev, ref.status, ref.cost)
    assert np.array_equal(p, p_ref)


# the band solver's one-rank path of round 5 (no right-hand-side copy, no back-correction of the interiors' columns: the separators'
# share of E^T C^-1 E as further rows of the Schur product, the step corrected for one vector) against the path of rounds 2-4
# (MVUS_PART_BACK=1, still what time shards run) and against the dense solve; interiors of 16 and of 32 control points
@pytest.mark.parametrize('cams,calib,knots', [(7, False, 40), (5, True, 64), (20, False, 150), (33, False, 90)])
def test_band_solver_without_back_correction(cams, calib, knots, monkeypatch):
    from mvus_amd import synth
    from mvus_amd.ba import BAHandle
    sc = synth.make_scene(cams, 120 * cams + 600, seed=300 + cams, rolling_shutter=True, num_knots=knots, opt_calib=calib)
    prob, x0 = mp.problem_from_scene(sc)
    lams = (1e-3, 0.7)

    def steps(env):
        for k in ('MVUS_PART_BACK', 'MVUS_DIRECT_RHS', 'MVUS_PART_LEN'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with BAHandle(prob) as h:
            h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
            out = [h.lm_step(lam) for lam in lams]
            ref = [_dense_step(prob, h, lam) for lam in lams] if not env else None
        return out, ref

    new, refs = steps({})
    again, _ = steps({})
    old, _ = steps({'MVUS_PART_BACK': '1'})
    copy, _ = steps({'MVUS_DIRECT_RHS': '0'})
    other_len, _ = steps({'MVUS_PART_LEN': '32' if prob.C * (3 + prob.P) <= 128 else '16'})
    for lam, p, a, o, c, l, r in zip(lams, new, again, old, copy, other_len, refs):
        scale = np.abs(r).max()
        assert np.array_equal(p, a), 'the solve is not reproducible run to run'
        print('lambda = %g: |p - dense| = %.2e, |p - back-corrected| = %.2e, |p - with the copy| = %.2e, |p - other interior length| = %.2e (relative)'
              % (lam, np.abs(p - r).max() / scale, np.abs(p - o).max() / scale, np.abs(p - c).max() / scale, np.abs(p - l).max() / scale))
        np.testing.assert_allclose(p, r, rtol=0, atol=1e-7 * scale)
        for q in (o, c, l):
            np.testing.assert_allclose(p, q, rtol=0, atol=1e-8 * scale)
