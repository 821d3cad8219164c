"""BASELINE.json configs at their full sizes on one MI355X (`-m gpu`), each against the oracle or through
size-independent properties:

  configs[1]  dataset4-shaped: 7 cams x ~100k detections, rolling shutter, motion_reg 'F'
  configs[4]  7 cams, opt_calib + rs_bounds + motion_reg 'KE' (full parameter vector)
  configs[3]  64 cams x 2M observations, 8 003 control points, the 576-unknown reduced camera system -- the
              configuration BASELINE shards over 8 GPUs, here whole on one (it fits: 675 MB of Jacobian)
(configs[0] and configs[2] are covered by tests/test_gpu_parity.py.)"""
import numpy as np
import pytest

from oracle import ba_oracle as orc
from mvus_amd import _lib, synth
from mvus_amd import problem as mp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def BAHandle():
    from mvus_amd.ba import BAHandle as H
    return H


@pytest.mark.parametrize('index', [1, 4])
def test_seven_camera_configs_vs_oracle(BAHandle, index):
    """Residual (detection + motion rows) and the integer outlier mask against the oracle on all ~100k detections, at the
    start and after a short LM solve; the solve's cost is re-evaluated by the oracle."""
    sc = synth.baseline_scene(index)
    prob, x0 = mp.problem_from_scene(sc)
    oprob, ox0 = orc.problem_from_scene(sc)
    assert prob.C == 7 and abs(prob.M - 100_000) < 5_000
    assert prob.motion_reg and prob.rs_free and (prob.opt_calib and prob.rs_bounds) == (index == 4)
    np.testing.assert_array_equal(x0, ox0)
    with BAHandle(prob) as h:
        f, fo = h.residual(x0), orc.residual(oprob, x0)
        assert f.shape == fo.shape and np.array_equal(f == 0, fo == 0)
        assert np.max(np.abs(f - fo) / np.maximum(1.0, np.abs(fo))) < 1e-9
        assert np.array_equal(h.outlier_mask(x0, 10.0), np.concatenate(orc.outlier_keep_mask(oprob, x0, 10.0)))
        r = h.solve(x0, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=8)
        assert r.cost < r.initial_cost
        fo2 = orc.residual(oprob, r.x)
        assert abs(0.5 * float(fo2 @ fo2) - r.cost) <= 1e-9 * r.cost
        assert np.max(np.abs(r.fun - fo2) / np.maximum(1.0, np.abs(fo2))) < 1e-9
        assert np.array_equal(h.outlier_mask(r.x, 10.0), np.concatenate(orc.outlier_keep_mask(oprob, r.x, 10.0)))
        if prob.rs_bounds:
            rs = r.x[2 * prob.C:3 * prob.C]
            assert np.all((rs >= 0) & (rs <= 1))
        # the reference-faithful solver on the same data: a few evaluations, cost re-evaluated by the oracle
        r2 = h.solve(x0, solver=_lib.SOLVER_TRF_LSMR, jac_mode=_lib.JAC_PATTERN, max_nfev=4)
        fo3 = orc.residual(oprob, r2.x)
        assert r2.cost < r2.initial_cost and abs(0.5 * float(fo3 @ fo3) - r2.cost) <= 1e-9 * r2.cost


def test_config3_on_one_gpu(BAHandle):
    """64 cams x 2M obs: operator identities, gradient of the assembled normal equations == J^T f, and the damped step of
    the whole solve chain (246 separators, 576 reduced unknowns = 18 Gauss-Jordan panels) against LAPACK."""
    from lm_reference import lapack_lm_step
    sc = synth.baseline_scene(3)
    prob, x0 = mp.problem_from_scene(sc)
    assert prob.C == 64 and abs(prob.M - 2_000_000) < 50_000 and int(prob.n_coef.sum()) > 7_500
    rng = np.random.default_rng(1)
    with BAHandle(prob) as h:
        f1 = h.residual(x0)
        # the detections against the oracle (the oracle evaluates whole cameras): every 8th camera and the last one, residual and
        # the integer outlier test of Scene.remove_outliers (common.py:709-713) on the same rows
        oprob = orc.problem_from_scene(sc)[0]
        alpha, beta, rs, cams, tck = orc.unpack_x(oprob, x0)
        keep_all = h.outlier_mask(x0, 10.0)
        for c in sorted(set(range(0, prob.C, 8)) | {prob.C - 1}):
            a, b = int(prob.det_offsets[c]), int(prob.det_offsets[c + 1])
            fo = orc.error_cam_each(oprob, c, alpha, beta, rs, cams[c], tck)
            assert np.max(np.abs(f1[2 * a:2 * b] - fo) / np.maximum(1.0, np.abs(fo))) < 1e-9
            assert np.array_equal(f1[2 * a:2 * b] == 0, fo == 0)
            assert np.array_equal(keep_all[a:b], np.sqrt(fo[:b - a] ** 2 + fo[b - a:] ** 2) < 10.0)
        v, u = rng.normal(size=h.n), rng.normal(size=h.m)
        f2, J, ctrl = h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        del J
        assert np.array_equal(f1, f2) and (ctrl >= 0).mean() > 0.9
        y, z = h.jv(v), h.jtu(u)
        assert abs(float(y @ u) - float(v @ z)) <= 1e-9 * np.linalg.norm(y) * np.linalg.norm(u)       # adjoint pair
        gg, A, band, cross = h.normal_equations()
        assert A.shape == (64, 9, 9) and cross.shape[0] == 64
        z_f = h.jtu(f1)
        np.testing.assert_allclose(gg, z_f, rtol=0, atol=1e-9 * np.abs(z_f).max())
        lam = 0.5
        p_gpu = h.lm_step(lam)
        p_ref = lapack_lm_step(prob, gg, A, band, cross, lam)
        np.testing.assert_allclose(p_gpu, p_ref, rtol=0, atol=1e-6 * np.abs(p_ref).max())
        del A, band, cross
        r = h.solve(x0, solver=_lib.SOLVER_LM_SCHUR, jac_mode=_lib.JAC_ANALYTIC, max_nfev=4, return_fun=False)
        assert r.cost < r.initial_cost
        keep = h.outlier_mask(x0, 10.0)
        off = prob.det_offsets
        ex = np.concatenate([f1[2 * a:2 * a + (b - a)] for a, b in zip(off[:-1], off[1:])])
        ey = np.concatenate([f1[2 * a + (b - a):2 * b] for a, b in zip(off[:-1], off[1:])])
        assert np.array_equal(keep, np.sqrt(ex ** 2 + ey ** 2) < 10.0)
