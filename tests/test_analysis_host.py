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


def test_align_gt_input_handling(tmp_path):
    scene, g = _flight('align_gt_2cam')
    assert compare_gt.align_gt(scene, 5.0, '', verbose=False) is None
    p = tmp_path / 'gt.txt'
    np.savetxt(p, np.asarray(g['gt']).T)                      # samples as rows: transposed on load
    out = compare_gt.align_gt(scene, float(g['f_gt']), str(p), verbose=False)
    np.testing.assert_allclose(out['align_param'], g['align_param'], rtol=1e-6)
    with pytest.raises(Exception):
        compare_gt.align_gt(scene, 5.0, np.zeros((5, 7)), verbose=False)


def test_traj_to_spline_and_back_match_the_reference():
    """Spline maintenance either side of BA (common.py:224-301) against the reference's own output
    (tests/golden/traj_spline.npz from tests/golden/make_golden_spline.py): same intervals, knots and coefficients from the
    smoothing loop, same resampled trajectories."""
    import os
    from golden_util import GOLDEN_DIR
    g = dict(np.load(os.path.join(GOLDEN_DIR, 'traj_spline.npz')))
    s = common.Scene()
    s.traj = g['traj'].copy()
    sp = s.traj_to_spline(smooth_factor=list(g['smooth_factor']))
    np.testing.assert_array_equal(sp['int'], g['interval'])
    assert len(sp['tck']) == int(g['n_int'])
    for i, tck in enumerate(sp['tck']):
        np.testing.assert_allclose(tck[0], g['knots_%d' % i], rtol=0, atol=1e-12)
        np.testing.assert_allclose(np.asarray(tck[1]), g['coefs_%d' % i], rtol=0, atol=1e-9)
    np.testing.assert_allclose(s.spline_to_traj(sampling_rate=1), g['traj_rate1'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(s.spline_to_traj(t=g['t_query']), g['traj_query'], rtol=0, atol=1e-9)
