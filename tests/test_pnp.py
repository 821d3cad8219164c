"""Scene.get_camera_pose (reference common.py:719-750): ``cv2.solvePnPRansac`` restated for the GPU (mvus_pnp_ransac).

OpenCV is not in this image and nothing of it lives under /root/reference: no reference output exists to compare with (parity
unpinned, said so in the header of oracle/pnp_oracle.py and in DESIGN.md).  What is checked instead:
* the device math compiled for the host (tests/hostcheck) against the oracle's restatement of the published model: projection
  with distortion, the direct-linear-transform pose, the Gauss-Newton terms (against finite differences), the rotation log;
* on the GPU: the pose against ground truth on synthetic correspondences with noise and gross outliers, the inlier flags
  against the ground-truth labelling, the refined pose against an independent minimiser of the same objective on the same
  inliers (scipy Levenberg-Marquardt, 1e-7), determinism, error behaviour, and Scene.get_camera_pose end to end.
"""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvus_amd import _lib                                  # noqa: E402
from oracle import pnp_oracle as po                        # noqa: E402

K = np.array([[1100.0, 0.0, 960.0], [0.0, 1080.0, 540.0], [0.0, 0.0, 1.0]])
KV = np.array([1100.0, 1080.0, 960.0, 540.0])


def _problem(seed, N=600, noise=0.5, outliers=0.1, dist=(-0.12, 0.03, 1e-3, -5e-4, 0.0)):
    rng = np.random.default_rng(seed)
    t = np.linspace(0.0, 600.0, N)
    X = np.vstack((10 * np.sin(t / 80), 10 * np.cos(t / 95), 30 + 3 * np.sin(t / 50))) + rng.normal(0, 0.05, (3, N))
    rvec = rng.normal(0, 0.4, 3)
    R = po.rodrigues(rvec)
    centre = np.array([3.0, -2.0, -15.0]) + rng.normal(0, 2.0, 3)
    tvec = -R @ centre
    d = np.array(dist)
    uv, depth = po.project(K, d, R, tvec, X)
    assert np.all(depth > 1.0)
    uv = uv + rng.normal(0, noise, uv.shape)
    bad = rng.random(N) < outliers
    uv[:, bad] += rng.uniform(30, 200, (2, int(bad.sum()))) * rng.choice([-1, 1], (2, int(bad.sum())))
    return X, uv, d, R, tvec, bad


def _angle(Ra, Rb):
    return np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


# ---- host build of the device math --------------------------------------------------------------------------------------

def test_projection_and_dlt_pose_on_the_host():
    from hostcheck_util import load
    lib = load()
    X, uv, d, R, t, _bad = _problem(1, noise=0.0, outliers=0.0)
    for i in (0, 17, 333):
        out = np.zeros(2)
        assert lib.hostcheck_pnp_project(_lib.dptr(KV), _lib.dptr(d), _lib.dptr(np.ascontiguousarray(R)), _lib.dptr(t), _lib.dptr(np.ascontiguousarray(X[:, i])), _lib.dptr(out))
        np.testing.assert_allclose(out, uv[:, i], rtol=0, atol=1e-9)
    xn = po.undistort(uv, K, d, iterations=40)
    rng = np.random.default_rng(2)
    m, s = X.mean(axis=1, keepdims=True), X.std()
    for _ in range(20):
        idx = rng.choice(X.shape[1], 6, replace=False)
        Xs = np.ascontiguousarray(((X[:, idx] - m) / s).T)
        xs = np.ascontiguousarray(xn[:, idx].T)
        Rh, th = np.zeros(9), np.zeros(3)
        ok = lib.hostcheck_pnp_dlt6(_lib.dptr(Xs), _lib.dptr(xs), _lib.dptr(Rh), _lib.dptr(th))
        Ro, to = po.dlt_pose(Xs.T, xs.T)
        assert ok
        np.testing.assert_allclose(Rh.reshape(3, 3), Ro, rtol=0, atol=1e-6)
        np.testing.assert_allclose(th, to, rtol=0, atol=1e-6 * max(1.0, np.abs(to).max()))
        # noise-free data: the pose is the true one (in the centred, scaled frame R is unchanged)
        assert _angle(Rh.reshape(3, 3), R) < 1e-4


def test_gauss_newton_terms_and_rotation_log_on_the_host():
    from hostcheck_util import load
    lib = load()
    X, uv, d, R, t, _bad = _problem(3, N=50)
    Rc = np.ascontiguousarray(R)
    for i in range(5):
        Xi = np.ascontiguousarray(X[:, i])
        acc = np.zeros(28)
        assert lib.hostcheck_pnp_point_normal(_lib.dptr(KV), _lib.dptr(d), _lib.dptr(Rc), _lib.dptr(t), _lib.dptr(Xi), float(uv[0, i]), float(uv[1, i]), _lib.dptr(acc))

        def res(delta):
            Rn = po.rodrigues(delta[:3]) @ R
            p, _ = po.project(K, d, Rn, t + delta[3:], Xi.reshape(3, 1))
            return np.ravel(p) - uv[:, i]
        J = np.zeros((2, 6))
        for a in range(6):
            e = np.zeros(6); e[a] = 1e-6
            J[:, a] = (res(e) - res(-e)) / 2e-6
        r0 = res(np.zeros(6))
        H = J.T @ J
        np.testing.assert_allclose(acc[:21], H[np.tril_indices(6)], rtol=1e-6, atol=1e-6 * np.abs(H).max())
        np.testing.assert_allclose(acc[21:27], J.T @ r0, rtol=1e-6, atol=1e-6 * np.abs(J.T @ r0).max())
        np.testing.assert_allclose(acc[27], r0 @ r0, rtol=1e-12)
    rng = np.random.default_rng(5)
    for r in [rng.normal(0, 1, 3) for _ in range(20)] + [np.array([np.pi - 1e-9, 0, 0]), np.array([0, 1e-10, 0]), np.zeros(3), np.array([2.2, -2.2, 0.3])]:
        Rm = np.ascontiguousarray(po.rodrigues(r))
        out = np.zeros(3)
        lib.hostcheck_rotation_to_rvec(_lib.dptr(Rm), _lib.dptr(out))
        np.testing.assert_allclose(po.rodrigues(out), Rm, rtol=0, atol=1e-8)
    idx = np.zeros(6, dtype=np.int64)
    lib.hostcheck_pnp_sample6(7, 3, 40, idx.ctypes.data_as(_lib.c_int64_p))
    assert len(set(idx.tolist())) == 6 and idx.min() >= 0 and idx.max() < 40


# ---- GPU ------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize('seed,outliers,dist', [(11, 0.1, (-0.12, 0.03, 1e-3, -5e-4, 0.0)), (12, 0.3, (0.0, 0.0, 0.0, 0.0, 0.0)),
                                                 (13, 0.02, (0.2, -0.1, 0.0, 0.0, 0.05))])
def test_gpu_pnp_ransac_recovers_the_pose(seed, outliers, dist):
    from mvus_amd.reconstruction.pnp import solve_pnp_ransac
    X, uv, d, R, t, bad = _problem(seed, outliers=outliers, dist=dist)
    ok, rvec, tvec, inl = solve_pnp_ransac(X.T, uv.T, K, d, reprojectionError=8.0)
    assert ok and rvec.shape == (3, 1) and tvec.shape == (3, 1) and inl.dtype == np.int32 and inl.shape[1] == 1
    Rg = po.rodrigues(np.ravel(rvec))
    assert _angle(Rg, R) < 0.05                                            # degrees; 0.5 px noise over 600 points
    assert np.linalg.norm(np.ravel(tvec) - t) < 2e-2 * np.linalg.norm(t)
    mask = np.zeros(X.shape[1], dtype=bool); mask[inl[:, 0]] = True
    assert np.array_equal(mask, ~bad)                                      # gross outliers are >= 30 px off, noise is 0.5 px
    # the refined pose minimises the reprojection error over those inliers: an independent minimiser lands on the same pose
    Ro, to, cost = po.refine(K, d, Rg, np.ravel(tvec), X[:, mask], uv[:, mask])
    assert _angle(Rg, Ro) < 1e-6
    np.testing.assert_allclose(np.ravel(tvec), to, rtol=0, atol=1e-7 * np.linalg.norm(to))
    proj, _ = po.project(K, d, Rg, np.ravel(tvec), X[:, mask])
    assert 0.5 * np.sum((proj - uv[:, mask]) ** 2) <= cost * (1 + 1e-9)
    # deterministic: the sampling is counter based
    again = solve_pnp_ransac(X.T, uv.T, K, d, reprojectionError=8.0)
    assert np.array_equal(again[1], rvec) and np.array_equal(again[2], tvec) and np.array_equal(again[3], inl)
    # OpenCV's array shapes are accepted too
    ocv = solve_pnp_ransac(X.T.reshape(-1, 1, 3), uv.T.reshape(-1, 1, 2), K, d.reshape(1, 5), reprojectionError=8.0)
    assert np.array_equal(ocv[1], rvec)


@pytest.mark.gpu
def test_gpu_pnp_ransac_error_paths_and_no_consensus():
    from mvus_amd.reconstruction.pnp import solve_pnp_ransac
    X, uv, d, R, t, _bad = _problem(21, N=100)
    with pytest.raises(ValueError):
        solve_pnp_ransac(X.T[:5], uv.T[:5], K, d)                          # fewer than six points
    with pytest.raises(ValueError):
        solve_pnp_ransac(X.T, uv.T[:50], K, d)
    Xn = X.copy(); Xn[0, 3] = np.nan
    with pytest.raises(ValueError):
        solve_pnp_ransac(Xn.T, uv.T, K, d)
    with pytest.raises(ValueError):
        solve_pnp_ransac(X.T, uv.T, K, d, reprojectionError=0.0)
    rng = np.random.default_rng(0)
    junk = rng.uniform(0, 1900, uv.shape)                                  # image points unrelated to the object points
    ok, rvec, tvec, inl = solve_pnp_ransac(X.T, junk.T, K, d, reprojectionError=0.05)
    assert ok is False and rvec is None and inl is None


@pytest.mark.gpu
def test_gpu_scene_get_camera_pose():
    """Scene.get_camera_pose end to end on a synthetic flight: a camera whose pose was lost gets it back from the trajectory
    spline and its own detections (time stamps through alpha, beta; 2 % gross outliers in the detections)."""
    from mvus_amd import synth
    from mvus_amd.reconstruction import common
    sc = synth.make_scene(3, 1500, seed=5, knot_spacing=15.0, perturb=0.0, distortion=True, rolling_shutter=True)
    s = common.Scene()
    s.numCam = sc.num_cam
    s.settings = dict(sc.settings)
    for c in sc.cameras:
        cam = common.Camera(K=c['K'].copy(), d=c['d'].copy(), R=c['R'].copy(), t=c['t'].copy(), fps=c['fps'], resolution=list(c['resolution']))
        cam.compose()
        s.addCamera(cam)
    for det in sc.detections:
        s.addDetection(det.copy())
    s.alpha, s.beta, s.rs = sc.alpha.copy(), sc.beta.copy(), sc.rs.copy()
    s.sequence = list(range(sc.num_cam))
    s.spline = {'tck': [[t.copy(), [c.copy() for c in cs], 3] for t, cs, _ in sc.tck], 'int': sc.interval.copy()}
    s.settings['undist_points'] = False          # raw detections + d go to the PnP, as the reference hands them to OpenCV (common.py:737-744)
    s.detection_to_global()
    cam = s.cameras[2]
    R_true, t_true = cam.R.copy(), cam.t.copy()
    cam.R, cam.t = np.eye(3), np.zeros(3)
    cam.compose()
    s.get_camera_pose(2, error=8)
    assert _angle(cam.R, R_true) < 0.1
    assert np.linalg.norm(cam.t - t_true) < 3e-2 * max(1.0, np.linalg.norm(t_true))
    np.testing.assert_allclose(cam.P, cam.K @ np.hstack((cam.R, cam.t.reshape(3, 1))), rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_gpu_incremental_loop_adds_a_camera():
    """The loop of the reference's main.py:44-83 for one added camera, every step on the GPU: BA on two cameras -> remove_outliers
    -> BA -> get_camera_pose of the third (its pose unknown) -> triangulate its detections into the trajectory -> BA on three.
    Ground truth decides: the third camera's pose, and the reprojection error of all three cameras after the last BA."""
    from mvus_amd import synth
    from mvus_amd.reconstruction import common
    sc = synth.make_scene(3, 3000, seed=9, knot_spacing=15.0, perturb=0.3)
    s = common.Scene()
    s.numCam = sc.num_cam
    s.settings = dict(sc.settings)
    s.settings.update(undist_points=False, sampling_rate=0.02, thres_triangulation=20)
    for c in sc.cameras:
        cam = common.Camera(K=c['K'].copy(), d=c['d'].copy(), R=c['R'].copy(), t=c['t'].copy(), fps=c['fps'], resolution=list(c['resolution']))
        cam.compose()
        s.addCamera(cam)
    for det in sc.detections:
        s.addDetection(det.copy())
    s.alpha, s.beta, s.rs = sc.alpha.copy(), sc.beta.copy(), sc.rs.copy()
    s.sequence = list(range(sc.num_cam))
    s.spline = {'tck': [[t.copy(), [c.copy() for c in cs], 3] for t, cs, _ in sc.tck], 'int': sc.interval.copy()}
    s.detection_to_global()
    truth = sc.truth
    R_true, t_true = truth['cameras'][2]['R'], truth['cameras'][2]['t']
    s.cameras[2].R = s.cameras[2].t = s.cameras[2].P = None              # the camera to be added: pose unknown
    s.sequence, s.find_order = [0, 1], True
    s.BA(2, max_iter=20)
    s.remove_outliers([0, 1], thres=10)
    s.BA(2, max_iter=20)
    s.select_most_overlap()
    assert s.sequence == [0, 1, 2]
    s.get_camera_pose(s.sequence[2], error=8)
    assert _angle(s.cameras[2].R, R_true) < 2.0           # the two-camera BA is free to drift in its gauge: truth only bounds the pose
    s.spline_to_traj(sampling_rate=1)
    n_before = s.traj.shape[1]
    s.triangulate(2, [0, 1], factor_t2s=s.settings['smooth_factor'], factor_s2t=s.settings['sampling_rate'], thres=s.settings['thres_triangulation'])
    assert s.traj.shape[1] >= n_before
    s.BA(3, max_iter=20)
    s.remove_outliers([0, 1, 2], thres=10)
    s.BA(3, max_iter=20)
    for c in range(3):
        err = s.error_cam(c, mode='each')
        m = err.size // 2
        dist = np.sqrt(err[:m] ** 2 + err[m:] ** 2)
        assert np.mean(dist[dist > 0]) < 1.5                               # 0.5 px detection noise per axis
    assert _angle(s.cameras[2].R, R_true) < 2.0


def test_dlt_pose_against_the_references_own_dlt():
    """The six-point pose of k_pnp_hypotheses (host build) against the REAL reference's ``epipolar.solve_PnP``
    (tests/golden/pnp_dlt.npz, make_golden_pnp.py) on exact correspondences: the same projection matrix up to scale."""
    from golden_util import GOLDEN_DIR
    from hostcheck_util import load
    lib = load()
    g = dict(np.load(os.path.join(GOLDEN_DIR, 'pnp_dlt.npz')))
    Kg = g['K']
    for X, x, P_ref in zip(g['X'], g['x'], g['P_ref']):
        xn = np.vstack(((x[0] - Kg[0, 2]) / Kg[0, 0], (x[1] - Kg[1, 2]) / Kg[1, 1]))
        Rh, th = np.zeros(9), np.zeros(3)
        assert lib.hostcheck_pnp_dlt6(_lib.dptr(np.ascontiguousarray(X.T)), _lib.dptr(np.ascontiguousarray(xn.T)), _lib.dptr(Rh), _lib.dptr(th))
        P = Kg @ np.hstack((Rh.reshape(3, 3), th.reshape(3, 1)))
        P = P / np.linalg.norm(P)
        np.testing.assert_allclose(P, P_ref, rtol=0, atol=1e-7)
