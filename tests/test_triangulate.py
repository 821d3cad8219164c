"""Triangulation step: reference epipolar.triangulate_matlab (epipolar.py:497-510) and
Scene.triangulate (common.py:754-815) against tests/golden/triangulate_3cam.npz, which
tests/golden/make_golden_triangulate.py produced by running the REAL reference."""
import os

import numpy as np
import pytest

from golden_util import GOLDEN_DIR
from oracle import triangulate_oracle as tri

X_RTOL = 1e-9       # relative to the point's norm: Jacobi SVD on the GPU vs LAPACK gesdd in the reference
ERR_ATOL = 1e-7     # px, reprojection distances


@pytest.fixture(scope='module')
def g():
    return dict(np.load(os.path.join(GOLDEN_DIR, 'triangulate_3cam.npz'), allow_pickle=False))


def _check_points(X, g):
    ref = g['fn_X']
    assert X.shape == ref.shape
    np.testing.assert_array_equal(X[3], 1.0)
    scale = np.linalg.norm(ref[:3], axis=0)
    assert np.max(np.linalg.norm(X[:3] - ref[:3], axis=0) / scale) < X_RTOL


def test_oracle_matches_the_reference(g):
    X = tri.triangulate_matlab(g['fn_x1'][1:], g['fn_x2'][1:], g['fn_P1'], g['fn_P2'])
    np.testing.assert_array_equal(X, g['fn_X'])                              # same LAPACK call: bit for bit
    np.testing.assert_allclose(tri.reprojection_error(g['fn_x1'][1:], tri.project(g['fn_P1'], X[:-1])), g['fn_err1'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(tri.reprojection_error(g['fn_x2'][1:], tri.project(g['fn_P2'], X[:-1])), g['fn_err2'], rtol=0, atol=1e-12)


def test_device_math_on_the_host_matches_the_reference(g):
    """csrc/triangulate.hip.h compiled with g++ (tests/hostcheck): the arithmetic the kernel runs."""
    from hostcheck_util import host_triangulate
    X, e1, e2 = host_triangulate(g['fn_x1'][1:], g['fn_x2'][1:], g['fn_P1'], g['fn_P2'])
    _check_points(X, g)
    np.testing.assert_allclose(e1, g['fn_err1'], rtol=0, atol=ERR_ATOL)
    np.testing.assert_allclose(e2, g['fn_err2'], rtol=0, atol=ERR_ATOL)
    # degenerate input: the same pixel in two identical cameras -> a point on the ray, still finite or cleanly non-finite, never a hang
    X2, _, _ = host_triangulate(g['fn_x1'][1:, :4], g['fn_x1'][1:, :4], g['fn_P1'], g['fn_P1'])
    assert X2.shape == (4, 4)


@pytest.mark.gpu
def test_gpu_triangulate_matches_the_reference(g):
    from mvus_amd.reconstruction import epipolar as ep
    X = ep.triangulate_matlab(g['fn_x1'][1:], g['fn_x2'][1:], g['fn_P1'], g['fn_P2'])
    _check_points(X, g)
    X2, e1, e2 = ep.triangulate_with_errors(np.vstack((g['fn_x1'][1:], np.ones(g['fn_x1'].shape[1]))),       # homogeneous rows are accepted
                                            g['fn_x2'][1:], g['fn_P1'], g['fn_P2'])
    np.testing.assert_array_equal(X2, X)
    np.testing.assert_allclose(e1, g['fn_err1'], rtol=0, atol=ERR_ATOL)
    np.testing.assert_allclose(e2, g['fn_err2'], rtol=0, atol=ERR_ATOL)
    # the threshold decision of Scene.triangulate (integer information): identical masks
    assert np.array_equal((e1 < 20) & (e2 < 20), (g['fn_err1'] < 20) & (g['fn_err2'] < 20))
    assert ep.triangulate_matlab(np.zeros((2, 0)), np.zeros((2, 0)), g['fn_P1'], g['fn_P2']).shape == (4, 0)  # empty input
    with pytest.raises(ValueError):
        ep.triangulate_matlab(g['fn_x1'][1:], g['fn_x2'][1:, :5], g['fn_P1'], g['fn_P2'])
    # a large batch: every lane independent, so tiling the input must tile the output bit for bit
    rep = 400
    Xb = ep.triangulate_matlab(np.tile(g['fn_x1'][1:], rep), np.tile(g['fn_x2'][1:], rep), g['fn_P1'], g['fn_P2'])
    assert Xb.shape[1] == rep * X.shape[1] and np.array_equal(Xb, np.tile(X, rep))


@pytest.mark.gpu
def test_scene_triangulate_matches_the_reference(g):
    """Scene.triangulate end to end: new points, appended trajectory and the refitted spline equal the reference's."""
    from golden_util import load_case      # noqa: F401  (same loader conventions)
    from mvus_amd.reconstruction import common
    C = int(g['num_cam'])
    off = g['det_offsets']
    s = common.Scene()
    s.numCam = C
    s.settings = dict(undist_points=bool(g['undist_points']), opt_calib=False, smooth_factor=[10, 20])
    for i in range(C):
        c = common.Camera(K=g['cam_K'][i].copy(), d=g['cam_d'][i].copy(), R=g['cam_R'][i].copy(), t=g['cam_t'][i].copy(),
                          fps=float(g['cam_fps'][i]), resolution=[float(v) for v in g['cam_res'][i]])
        c.compose()
        s.addCamera(c)
        s.addDetection(g['detections'][:, off[i]:off[i + 1]].copy())
    s.alpha, s.beta, s.rs = g['alpha'].copy(), g['beta'].copy(), g['rs'].copy()
    s.sequence = list(range(C))
    s.detection_to_global()
    s.traj = g['sc_traj_in'].copy()
    s.traj_to_spline(smooth_factor=[10, 20])
    np.testing.assert_array_equal(s.spline['int'], g['sc_int_before'])
    X_new = s.triangulate(2, [0, 1], factor_t2s=[10, 20], factor_s2t=0.02, thres=float(g['sc_thres']))
    ref = g['sc_X_new']
    assert X_new.shape == ref.shape                                              # same points pass the threshold
    np.testing.assert_array_equal(X_new[0], ref[0])                              # same timestamps
    assert np.max(np.linalg.norm(X_new[1:] - ref[1:], axis=0) / np.linalg.norm(ref[1:], axis=0)) < X_RTOL
    assert tuple(s.traj.shape) == tuple(g['sc_traj_out_shape'])
    np.testing.assert_allclose(s.traj[:, ::40], g['sc_traj_out_sub'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(s.traj.sum(axis=1), g['sc_traj_out_sum'], rtol=1e-10)
    np.testing.assert_allclose(s.spline['int'], g['sc_int_after'], rtol=0, atol=1e-12)
    # The refit is FITPACK's smoothing spline with ADAPTIVE knot placement inside the reference's knot-density loop
    # (common.py:224-270): whether one more knot is inserted hinges on a threshold crossing of the residual sum, so inputs
    # that agree to 1e-9 (Jacobi SVD here, LAPACK there) can end with a few knots more or fewer (measured: 70 vs 64).  The
    # curve is what is compared: both splines evaluated on the same grid.
    from scipy import interpolate
    knots = np.concatenate([t[0] for t in s.spline['tck']])
    assert abs(knots.size - g['sc_knots_after'].size) <= 0.2 * g['sc_knots_after'].size
    koff = g['sc_knot_offsets_after']
    pos = 0
    for k in range(koff.size - 1):
        t_ref = g['sc_knots_after'][koff[k]:koff[k + 1]]
        n = t_ref.size - 4
        c_ref = g['sc_coefs_after'][pos:pos + 3 * n].reshape(3, n)
        pos += 3 * n
        grid = np.linspace(s.spline['int'][0, k], s.spline['int'][1, k], 4000)
        mine = np.asarray(interpolate.splev(grid, s.spline['tck'][k]))
        theirs = np.asarray(interpolate.splev(grid, [t_ref, list(c_ref), 3]))
        dev = np.linalg.norm(mine - theirs, axis=0)
        # measured: mean 6e-3, max 0.7 (one stretch of noisy new points where the knots fall differently), against an RMS
        # distance of the fitted data to either curve of 2.7e-2 -- the two fits differ by less than they differ from the data
        assert dev.mean() < 2e-2 and np.quantile(dev, 0.9) < 5e-2, (dev.mean(), np.quantile(dev, 0.9), dev.max())
