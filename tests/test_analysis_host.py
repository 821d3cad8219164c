"""Ground-truth alignment report against the reference's own output
(tests/golden/align_gt_2cam.npz, made by tests/golden/make_golden_align.py from the real analysis/compare_gt.py)."""
import numpy as np
import pytest

from golden_util import load_case
from mvus_amd.analysis import compare_gt
from mvus_amd.reconstruction import common


def _flight(name):
    """The drop-in Scene (mvus_amd.reconstruction.common) of a golden case."""
    scene, g = load_case(name)
    s = common.Scene()
    s.numCam = scene.num_cam
    s.settings = dict(scene.settings)
    for cam in scene.cameras:
        c = common.Camera(K=cam['K'].copy(), d=cam['d'].copy(), R=cam['R'].copy(), t=cam['t'].copy(), fps=cam['fps'],
                          resolution=list(cam['resolution']))
        c.compose()
        s.addCamera(c)
    s.spline = {'tck': [[t.copy(), [c.copy() for c in cs], 3] for t, cs, _ in scene.tck], 'int': scene.interval.copy()}
    s.settings['ref_cam'] = 0
    return s, g


def test_similarity_fit_recovers_a_known_transform():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(1)
    p = rng.normal(size=(3, 40))
    R = Rotation.from_rotvec([-0.7, 0.2, 0.4]).as_matrix()
    q = 1.8 * R @ p + np.array([[3.0], [-1.0], [0.5]])
    M = compare_gt.similarity_from_points(p, q)
    np.testing.assert_allclose(M[:3, :3], 1.8 * R, atol=1e-12)
    np.testing.assert_allclose(M[:3, 3], [3.0, -1.0, 0.5], atol=1e-12)
    np.testing.assert_allclose(compare_gt.error_M(M, np.vstack((p, q))), 0.0, atol=1e-12)
    with pytest.raises(ValueError):
        compare_gt.similarity_from_points(p[:, :2], q[:, :2])


@pytest.mark.gpu            # Scene.spline_to_traj evaluates on the GPU
def test_align_gt_matches_the_reference():
    scene, g = _flight('align_gt_2cam')
    out = compare_gt.align_gt(scene, float(g['f_gt']), np.asarray(g['gt']), verbose=False)
    np.testing.assert_allclose(out['align_param'], g['align_param'], rtol=1e-7)
    np.testing.assert_allclose(out['tran_matrix'], g['tran_matrix'], rtol=0, atol=1e-6 * np.abs(g['tran_matrix']).max())
    assert out['error'].size == g['error'].size
    np.testing.assert_allclose(out['error'], g['error'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(np.mean(out['error']), g['error_mean'], rtol=1e-6)
    # and it found the synthetic clock: one ground-truth sample is alpha frames, first one at beta
    assert abs(out['align_param'][0] - g['alpha_true']) < 5e-3 and abs(out['align_param'][1] - g['beta_true']) < 0.5
    assert np.median(out['error']) < 0.03                     # 1 cm noise on the synthetic ground truth


@pytest.mark.gpu
def test_align_gt_input_handling(tmp_path):
    scene, g = _flight('align_gt_2cam')
    assert compare_gt.align_gt(scene, 5.0, '', verbose=False) is None
    p = tmp_path / 'gt.txt'
    np.savetxt(p, np.asarray(g['gt']).T)                      # samples as rows: transposed on load
    out = compare_gt.align_gt(scene, float(g['f_gt']), str(p), verbose=False)
    np.testing.assert_allclose(out['align_param'], g['align_param'], rtol=1e-6)
    with pytest.raises(Exception):
        compare_gt.align_gt(scene, 5.0, np.zeros((5, 7)), verbose=False)


@pytest.mark.gpu
def test_spline_to_traj_matches_the_reference():
    """Scene.spline_to_traj (common.py:273-301) with the evaluation on the GPU, against the reference's resampled
    trajectories (constant rate and at given timestamps, two intervals with a gap) and against scipy's splev."""
    import os
    from scipy import interpolate
    from golden_util import GOLDEN_DIR
    from mvus_amd import spline
    g = dict(np.load(os.path.join(GOLDEN_DIR, 'traj_spline.npz')))
    s = common.Scene()
    s.settings = {}
    s.spline = {'tck': [[g['knots_%d' % i], list(g['coefs_%d' % i]), 3] for i in range(int(g['n_int']))], 'int': g['interval']}
    np.testing.assert_allclose(s.spline_to_traj(sampling_rate=1), g['traj_rate1'], rtol=0, atol=1e-11)
    np.testing.assert_allclose(s.spline_to_traj(t=g['t_query']), g['traj_query'], rtol=0, atol=1e-11)
    # closed interval ends, timestamps outside every interval, an empty query
    iv = g['interval']
    t = np.array([iv[0, 0] - 1.0, iv[0, 0], iv[1, 0], 0.5 * (iv[1, 0] + iv[0, 1]), iv[0, 1], iv[1, 1], iv[1, 1] + 1e-9])
    X, which = spline.evaluate(s.spline['tck'], iv, t)
    np.testing.assert_array_equal(which, [-1, 0, 0, -1, 1, 1, -1])
    for k in (1, 2, 4, 5):
        ref = np.asarray(interpolate.splev(t[k], s.spline['tck'][which[k]]))
        np.testing.assert_allclose(X[:, k], ref, rtol=0, atol=1e-11)
    assert spline.evaluate(s.spline['tck'], iv, np.zeros(0))[0].shape == (3, 0)
    assert s.spline_to_traj(t=np.array([iv[0, 0] - 5.0])).shape == (4, 0)


@pytest.mark.gpu
def test_lsq_fit_on_fixed_knots_matches_scipy():
    """mvus_spline_lsq: least-squares coefficients on the knots FITPACK placed, against scipy.interpolate.make_lsq_spline."""
    import os
    from scipy import interpolate
    from golden_util import GOLDEN_DIR
    from mvus_amd import spline
    g = dict(np.load(os.path.join(GOLDEN_DIR, 'traj_spline.npz')))
    traj = g['traj']
    for i in range(int(g['n_int'])):
        knots = g['knots_%d' % i]
        m = (traj[0] >= g['interval'][0, i]) & (traj[0] <= g['interval'][1, i])
        t, X = traj[0, m], traj[1:, m]
        c = spline.lsq_fit(knots, t, X)
        ref = interpolate.make_lsq_spline(t, X.T, knots, k=3).c.T
        np.testing.assert_allclose(np.asarray(c), ref, rtol=0, atol=1e-9 * np.abs(ref).max())
        # the least-squares fit is at least as close to the data as FITPACK's smoothing spline on the same knots
        fit = np.asarray(interpolate.splev(t, [knots, c, 3]))
        smooth = np.asarray(interpolate.splev(t, [knots, list(g['coefs_%d' % i]), 3]))
        assert np.sum((fit - X) ** 2) <= np.sum((smooth - X) ** 2) * (1 + 1e-9)
    with pytest.raises(ValueError):
        spline.lsq_fit(knots, t[:5] + 1e6, X[:, :5])                  # data outside the knot interval


@pytest.mark.gpu            # all_detect_to_traj resamples the spline (Scene.spline_to_traj, GPU)
@pytest.mark.parametrize('name', ['rs_F_2int_3cam', 'calib_KE_bounds_3cam'])
def test_all_detect_to_traj_vs_reference(name):
    """Scene.all_detect_to_traj (common.py:887-947) at the state the reference's first BA left behind: global_traj,
    global_detections, frame_id_all, global_time_stamps_all, traj -- the attributes of the output pickle -- equal to the
    reference's."""
    from golden_util import load_case
    from test_gpu_scene import build_scene
    from mvus_amd import problem as mp
    scene, g = load_case(name)
    s = build_scene(scene)
    prob, _ = mp.problem_from_scene(scene)
    alpha, beta, rs, cams, coefs = mp.unpack_x(prob, g['ba10_x'])
    s.alpha, s.beta, s.rs = alpha, beta, rs
    for k, c in enumerate(s.cameras):
        if prob.opt_calib:
            c.K, c.d = cams[k]['K'], cams[k]['d']
        c.R, c.t = cams[k]['R'], cams[k]['t']
        c.compose()
    for i, c in enumerate(coefs):
        s.spline['tck'][i][1] = c
    s.all_detect_to_traj(s.sequence[:s.numCam])
    np.testing.assert_array_equal(s.frame_id_all, g['adt_frame_id_all'])
    np.testing.assert_allclose(s.global_time_stamps_all, g['adt_global_time_stamps_all'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(s.global_detections, g['adt_global_detections'], rtol=0, atol=1e-9)
    assert s.global_traj.shape == g['adt_global_traj'].shape and s.traj.shape == g['adt_traj'].shape
    np.testing.assert_array_equal(s.global_traj[:3], g['adt_global_traj'][:3])          # order index, camera id, frame id: integers
    np.testing.assert_allclose(s.global_traj[3:], g['adt_global_traj'][3:], rtol=0, atol=1e-8)
    np.testing.assert_allclose(s.traj, g['adt_traj'], rtol=0, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['rs_F_2int_3cam', 'calib_KE_bounds_3cam'])
def test_error_motion_is_the_tail_of_the_reference_residual(name):
    """Scene.error_motion(motion_reg=True) (common.py:362-424) at the fixture's x0: the values the reference's error_BA
    appends to its residual vector (common.py:462-467; golden `f_x0`, rows 2M..), F and KE, one per sample of
    spline_to_traj(), zeros at the interval ends included."""
    from test_gpu_scene import build_scene
    scene, g = load_case(name)
    s = build_scene(scene)
    st = scene.settings
    M = sum(d.shape[1] for d in scene.detections)
    ref = g['f_x0'][2 * M:]
    em = s.error_motion(list(range(s.numCam)), motion_weights=st['motion_weights'], motion_reg=True)
    assert em.shape == ref.shape == (s.traj.shape[1],)
    assert np.array_equal(em == 0, ref == 0)
    np.testing.assert_allclose(em, ref, rtol=1e-9, atol=1e-9)
    with pytest.raises(NotImplementedError):
        s.error_motion([0], motion_prior=True)
    with pytest.raises(ValueError):
        s.error_motion([0])
