"""CPU check of the arithmetic the HIP kernels run (mvus_amd/csrc/ba_math.h, compiled for the host
by tests/hostcheck) against the oracle: residuals, visibility, analytic Jacobian vs central differences."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import ba_oracle as orc
from golden_util import CASES, load_case
from mvus_amd import problem as mp

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def hostlib():
    import hostcheck_util
    return hostcheck_util.load()           # ONE build recipe for tests/hostcheck/libhostcheck.so (both sources; rebuilt when a header is newer)


def host_eval(lib, prob, x):
    M, NS = prob.M, 3 + prob.P + 12
    ex, ey = np.zeros(M), np.zeros(M)
    ctrl = np.zeros(M, dtype=np.int32)
    J = np.zeros((M, 2, NS))
    c = lambda a, t: np.ascontiguousarray(a, dtype=t)
    arrs = dict(det_off=c(prob.det_offsets, np.int64), frame=c(prob.frame, np.float64), u=c(prob.u_raw, np.float64),
                v=c(prob.v_raw, np.float64), H=c(prob.img_height, np.float64), K=c(prob.K, np.float64),
                d=c(prob.dist, np.float64), i0=c(prob.interval[0], np.float64), i1=c(prob.interval[1], np.float64),
                knots=c(prob.knots, np.float64), koff=c(prob.knot_offsets, np.int32), coff=c(prob.ctrl_offsets, np.int32),
                xoff=c(prob.spline_x_offsets, np.int32), x=c(x, np.float64))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.hostcheck_eval(ctypes.c_int(prob.C), ctypes.c_int(prob.opt_calib), ctypes.c_int(prob.undist_points),
                       ctypes.c_int(prob.rs_free), p(arrs['det_off']), p(arrs['frame']), p(arrs['u']), p(arrs['v']),
                       p(arrs['H']), p(arrs['K']), p(arrs['d']), ctypes.c_int(prob.S), p(arrs['i0']), p(arrs['i1']),
                       p(arrs['knots']), p(arrs['koff']), p(arrs['coff']), p(arrs['xoff']), p(arrs['x']),
                       p(ex), p(ey), p(ctrl), p(J))
    return ex, ey, ctrl, J


def to_reference_rows(prob, ex, ey):
    out = []
    for c in range(prob.C):
        a, b = prob.det_offsets[c], prob.det_offsets[c + 1]
        out += [ex[a:b], ey[a:b]]
    return np.concatenate(out)


def slot_columns(prob, c, ctrl):
    """x-index of every Jacobian slot of an observation of camera c whose first control point is ctrl."""
    C, P = prob.C, prob.P
    cols = [c, C + c, 2 * C + c] + list(range(3 * C + c * P, 3 * C + (c + 1) * P))
    coff = prob.ctrl_offsets
    s = int(np.searchsorted(coff, ctrl, side='right') - 1)
    n = int(prob.n_coef[s])
    j = ctrl - int(coff[s])
    for q in range(4):
        for d in range(3):
            cols.append(int(prob.spline_x_offsets[s]) + d * n + j + q)
    return np.array(cols)


@pytest.mark.parametrize('name', CASES)
def test_host_residual_matches_oracle_and_golden(hostlib, name):
    scene, g = load_case(name)
    prob, x0 = mp.problem_from_scene(scene)
    np.testing.assert_allclose(x0, g['x0'], rtol=0, atol=1e-12)
    for x, fref in ((g['x0'], g['f_x0']), (g['x0'] + g['delta'], g['f_x0_delta'])):
        ex, ey, ctrl, _ = host_eval(hostlib, prob, x)
        f = to_reference_rows(prob, ex, ey)
        M2 = 2 * prob.M
        np.testing.assert_allclose(f, fref[:M2], rtol=0, atol=1e-9)
        assert np.array_equal(f == 0, fref[:M2] == 0)
        assert np.array_equal(ctrl >= 0, ex != 0)


@pytest.mark.parametrize('name', CASES)
def test_host_analytic_jacobian_vs_central_differences(hostlib, name):
    scene, g = load_case(name)
    prob, _ = mp.problem_from_scene(scene)
    oprob, _ = orc.problem_from_scene(scene)
    oprob.motion_reg = False
    x = g['x0'] + g['delta']
    ex, ey, ctrl, J = host_eval(hostlib, prob, x)
    Jfd = orc.numeric_jacobian(oprob, x, rel=1e-6)
    n = x.size
    Jdense = np.zeros((2 * prob.M, n))
    for c in range(prob.C):
        a, b = int(prob.det_offsets[c]), int(prob.det_offsets[c + 1])
        for i in range(a, b):
            if ctrl[i] < 0:
                continue
            cols = slot_columns(prob, c, int(ctrl[i]))
            Jdense[2 * a + (i - a), cols] = J[i, 0]
            Jdense[2 * a + (b - a) + (i - a), cols] = J[i, 1]
    if not prob.rs_free:
        Jfd[:, 2 * prob.C:3 * prob.C] = 0.0          # rs column is not a free parameter when rs=False
    # central differences are meaningless across the |r| kink and across a visibility flip: skip rows
    # whose residual is within reach of the step, and rows next to an interval boundary
    f = orc.residual(oprob, x)
    ok = np.abs(f) > 0.05
    alpha, beta, rs, cams, tck = orc.unpack_x(oprob, x)
    near_edge = []
    for c in range(prob.C):
        tau = orc.detection_to_global(oprob, c, alpha, beta, rs, cams[c])[0]
        ne = (np.abs(tau[:, None] - oprob.interval.reshape(1, -1)) < 0.05).any(axis=1)
        near_edge += [ne, ne]
    ok &= ~np.concatenate(near_edge)
    assert ok.sum() > 0.8 * np.count_nonzero(f)
    scale = np.maximum(np.abs(Jfd[ok]).max(axis=0), 1e-12)
    err = np.abs(Jdense[ok] - Jfd[ok]) / scale
    assert err.max() < 2e-5, (err.max(), np.unravel_index(err.argmax(), err.shape))
    # rows of invisible detections carry no derivative at all
    assert not Jdense[f == 0].any()


def test_undistort_negative_icdist_resets_the_point():
    """cv2.undistortPoints of OpenCV >= 4.1.1 (behind Camera.undist_point, common.py:1147-1157) leaves a point at its normalised
    start when 1 / (1 + k1 r^2 + k2 r^4 + k3 r^6) turns negative during the fixed-point iteration (cvUndistortPointsInternal,
    regression_14583).  The device math, the oracle and the golden generator's shim restate that: a camera with k1 = -0.9 and
    detections far from the principal point -- host build of the device math == oracle, and the affected observed pixels equal the
    raw ones."""
    from hostcheck_util import HostHandle
    from mvus_amd import synth
    sc = synth.make_scene(2, 600, seed=71, distortion=True)
    sc.cameras[0]['d'] = np.array([-0.9, 0.0, 0.0, 0.0, 0.0])
    K = sc.cameras[0]['K']
    d0 = sc.detections[0]
    d0[1] = K[0, 2] + 1.6 * K[0, 0] * np.sign(d0[1] - K[0, 2] + 0.5)      # |x0| = 1.6: 1 - 0.9 r^2 < 0
    prob, x0 = mp.problem_from_scene(sc)
    oprob, ox0 = orc.problem_from_scene(sc)
    und = orc.undist_point(d0[1:3], K, sc.cameras[0]['d'])
    assert np.allclose(und, d0[1:3], rtol=0, atol=1e-9)            # reset to the start: K * normalised(raw) = raw
    f_host = HostHandle(prob).residual(x0)
    f_orc = orc.residual(oprob, ox0)
    assert np.all(np.isfinite(f_host)) and np.max(np.abs(f_host - f_orc)) < 1e-9
