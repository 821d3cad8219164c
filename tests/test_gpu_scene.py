"""Scene-level drop-in test on the GPU: the call sequence of the reference's main.py (BA -> remove_outliers -> BA,
main.py:49-62) through mvus_amd.reconstruction.common, against the golden vectors of the same sequence."""
import numpy as np
import pytest

from golden_util import CASES, load_case
from mvus_amd import _lib
from mvus_amd.reconstruction import common

pytestmark = pytest.mark.gpu


def build_scene(scene):
    s = common.Scene()
    s.numCam = scene.num_cam
    s.settings = dict(scene.settings)
    for cam in scene.cameras:
        c = common.Camera(K=cam['K'].copy(), d=cam['d'].copy(), R=cam['R'].copy(), t=cam['t'].copy(), fps=cam['fps'],
                          resolution=list(cam['resolution']))
        c.compose()
        s.addCamera(c)
    for det in scene.detections:
        s.addDetection(det.copy())
    s.alpha, s.beta, s.rs = scene.alpha.copy(), scene.beta.copy(), scene.rs.copy()
    s.sequence = list(range(scene.num_cam))
    s.spline = {'tck': [[t.copy(), [c.copy() for c in cs], 3] for t, cs, _ in scene.tck], 'int': scene.interval.copy()}
    s.detection_to_global()
    return s


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('mode', ['default', 'lm'])
def test_ba_outliers_ba_sequence(name, mode):
    """main.py:49-62 through the drop-in Scene: 'default' = settings untouched = the reference's algorithm (TRF + LSMR over
    grouped 2-point differences); 'lm' = the fast solver, opted into with settings['ba_solver'] = 'lm'."""
    scene, g = load_case(name)
    st = scene.settings
    s = build_scene(scene)
    assert 'ba_solver' not in s.settings and 'ba_jacobian' not in s.settings
    if mode == 'lm':
        s.settings.update(ba_solver='lm')
    C = s.numCam
    kw = dict(rs=st['rolling_shutter'], motion_reg=st['motion_reg'], motion_weights=st['motion_weights'], rs_bounds=st['rs_bounds'])
    before = np.array([np.mean(s.error_cam(i)) for i in range(C)])
    np.testing.assert_allclose(before, g['mean_err_before'], rtol=0, atol=1e-9)      # same numbers main.py:46 prints
    res = s.BA(C, **kw)
    handle = s._ba_handle
    assert handle is not None and handle.h
    assert res.nfev == int(g['ba10_nfev'])
    loose = name == 'calib_KE_bounds_3cam'            # the scene whose 10-evaluation iterate the reference itself does not reproduce
    if mode == 'default':
        assert abs(res.cost - float(g['ba10_cost'])) < (2e-2 if loose else 5e-3) * float(g['ba10_cost'])
    elif not loose:                                   # (LM's first ten evaluations on the ill-posed calibration scene go nowhere useful)
        assert res.cost < float(g['ba10_cost']) * (1 + 5e-3)
    n_before = sum(d.shape[1] for d in s.detections)
    s.remove_outliers(s.sequence[:C], thres=st['thres_outlier'])
    removed = n_before - sum(d.shape[1] for d in s.detections)
    ref_removed = int((g['outlier_keep'] == 0).sum())
    # Both runs stop UNCONVERGED after 10 evaluations (the decidable comparison is test_converged_second_ba_through_the_scene below),
    # so the bars are the values measured on MI355X plus the chaos margin of an unconverged iterate, per scene.  Removed here /
    # by the reference -- default (the reference's algorithm): 123/126, 40/40, 87/78, 60/58; lm (another optimiser, another
    # 10-evaluation iterate): 156/126, 40/40, 90/78, 64/58
    removed_tol = {'c1_pinhole_2cam': 5, 'rs_F_2int_3cam': 1, 'calib_KE_bounds_3cam': 14, 'dist_fixed_2cam': 4}[name]
    if mode == 'default':
        assert abs(removed - ref_removed) <= removed_tol
    elif not loose:
        # (on the ill-posed calibration scene LM drives k3 to ~5e4 within ten evaluations: 1 / (1 + k1 r^2 + ...) turns negative for part
        # of the detections, OpenCV's early-out -- restated since round 4 -- resets those points, and the count of "outliers" at such an
        # iterate means nothing: 90 before the early-out was restated, 29 with it, 78 by the reference at ITS tenth evaluation)
        assert abs(removed - ref_removed) <= max(3, 0.4 * ref_removed)
    else:                                             # ... but it stays a small part of the scene (the threshold still means something)
        assert removed <= max(4 * ref_removed, 0.1 * n_before)
    assert s._ba_handle is handle and handle.M == sum(d.shape[1] for d in s.detections)    # filtered in place on the GPU
    res2 = s.BA(C, **kw)
    assert s._ba_handle is handle                                                            # no new handle, no re-upload
    import pickle
    assert pickle.loads(pickle.dumps(s))._ba_handle is None                                  # the output pickle carries data only
    rmse = np.sqrt(np.mean(np.concatenate([s.error_cam(i, 'dist') for i in range(C)]) ** 2))
    print('%s %s: removed %d (ref %d), second BA cost %.6g (ref %.6g), rmse %.6f (ref %.6f)'
          % (name, mode, removed, ref_removed, res2.cost, float(g['ba2_10_cost']), rmse, float(g['ba2_10_rmse'])))
    # Final answer of the pipeline after 10 + 10 evaluations, on (slightly) different inlier sets.  With motion_reg the cost
    # trades reprojection error against the heavily weighted motion term, so the RMSE bound is one-sided and loose there.
    assert res2.cost < res2.initial_cost
    # measured, second-BA cost / reference and RMSE - reference: default 0.9998 / -6e-4, 1.0001 / +1e-3, 1.119 / +0.13 (the ill-posed
    # calibration scene, whose 10-evaluation iterate the reference itself does not reproduce), 1.062 / +0.019; lm 0.94, 0.93, 1.07, 0.99
    cost_ratio = {'c1_pinhole_2cam': 1.005, 'rs_F_2int_3cam': 1.005, 'calib_KE_bounds_3cam': 1.25, 'dist_fixed_2cam': 1.1}[name]
    rmse_margin = {'c1_pinhole_2cam': 5e-3, 'rs_F_2int_3cam': 5e-3, 'calib_KE_bounds_3cam': 0.25, 'dist_fixed_2cam': 0.04}[name]
    if mode == 'default':
        assert res2.cost < float(g['ba2_10_cost']) * cost_ratio
        assert rmse < float(g['ba2_10_rmse']) + rmse_margin
    else:
        assert res2.cost < float(g['ba2_10_cost']) * 1.15
        assert rmse < float(g['ba2_10_rmse']) * (1.15 if st['motion_reg'] else 1.02) + 1e-3
    assert np.all(np.isfinite(s.alpha)) and len(s.detections_global) == C
    if st['motion_reg']:
        assert s.global_traj.shape[0] == 7 and s.traj.shape[0] == 4                 # attributes the pickle carries
    if st['rs_bounds']:
        assert np.all((s.rs >= 0) & (s.rs <= 1))


def scene_at_second_ba(name):
    """The drop-in Scene in the state the reference's pipeline is in when it calls its second BA (main.py:59): inliers of
    its first outlier pass, parameters of its first BA."""
    from mvus_amd import problem as mp
    from test_fd_mode_host import filtered_case
    scene, g = filtered_case(name)
    s = build_scene(scene)
    prob, _ = mp.problem_from_scene(scene)
    alpha, beta, rs, cams, coefs = mp.unpack_x(prob, g['ba2_200_x0'])
    s.alpha, s.beta, s.rs = alpha, beta, rs
    for k, c in enumerate(s.cameras):
        if prob.opt_calib:
            c.K, c.d = cams[k]['K'], cams[k]['d']
        c.R, c.t = cams[k]['R'], cams[k]['t']
        c.compose()
    for i, c in enumerate(coefs):
        s.spline['tck'][i][1] = c
    s.detection_to_global()
    return s, scene, g


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('mode', ['default', 'lm'])
def test_converged_second_ba_through_the_scene(name, mode):
    """Scene.BA(max_iter=200) from the reference's own start of its second BA: by default (the reference's algorithm over the
    reference's matrix, passed like the reference passes it to least_squares) the final RMSE -- computed by Scene.error_cam
    like main.py does -- is the reference's within SPREAD_FACTOR x its own reproducibility, the scene state BA leaves behind
    (alpha, beta, rs, cameras, spline: what common.py:672-695 writes back from res.x) is the reference's in gauge-invariant
    terms, and remove_outliers then removes exactly the detections the reference removes; the 'lm' solver ends at a lower
    value of the same objective."""
    import gauge
    from golden_util import reference_spread
    from oracle import ba_oracle as orc
    from mvus_amd import problem as mp
    from test_fd_mode_host import golden_matrix
    from test_gpu_parity import SPREAD_FACTOR
    s, scene, g = scene_at_second_ba(name)
    st = scene.settings
    C = s.numCam
    kw = dict(rs=st['rolling_shutter'], motion_reg=st['motion_reg'], motion_weights=st['motion_weights'], rs_bounds=st['rs_bounds'])
    if mode == 'default':
        res = s.BA(C, max_iter=200, jac_sparsity=golden_matrix(g, second=True), **kw)
    else:
        s.settings.update(ba_solver='lm')
        res = s.BA(C, max_iter=200, **kw)
    # the state Scene.BA wrote back, re-packed: must be res.x (the unpack / pack of common.py:672-695 loses nothing)
    prob, _ = mp.problem_from_scene(scene)
    x_state = mp.pack_x(prob, s.alpha, s.beta, s.rs, s.cameras, s.spline['tck'])
    Cn, P = prob.C, prob.P
    np.testing.assert_array_equal(x_state[:3 * Cn], res.x[:3 * Cn])
    np.testing.assert_array_equal(x_state[Cn * (3 + P):], res.x[Cn * (3 + P):])
    _, _, _, cam_states, _ = mp.unpack_x(prob, res.x)
    for cam, stt in zip(s.cameras, cam_states):                           # (a rotation vector beyond pi re-packs to its twin: compare R)
        np.testing.assert_array_equal(cam.R, stt['R'])
        np.testing.assert_array_equal(cam.t, stt['t'])
    rmse = np.sqrt(np.mean(np.concatenate([s.error_cam(i, 'dist') for i in range(C)]) ** 2))
    n_before = sum(d.shape[1] for d in s.detections)
    frames = [d[0].copy() for d in s.detections]
    s.remove_outliers(s.sequence[:C], thres=st['thres_outlier'])
    keep = np.concatenate([np.isin(fb, d[0]) for fb, d in zip(frames, s.detections)]).astype(np.uint8)
    print('%s %s: rmse %.6f (ref %.6f), cost %.6g (ref %.6g), status %d, removed %d (ref %d)'
          % (name, mode, rmse, float(g['ba2_200_rmse']), res.cost, float(g['ba2_200_cost']), res.status,
             n_before - int(keep.sum()), int((g['ba2_200_keep'] == 0).sum())))
    if mode == 'default':
        oprob, _ = orc.problem_from_scene(scene)
        spread = reference_spread(oprob, name, g['ba2_200_x'])
        c = gauge.compare(oprob, g['ba2_200_x'], x_state)
        assert res.status == int(g['ba2_200_status'])
        assert abs(rmse - float(g['ba2_200_rmse'])) <= SPREAD_FACTOR * spread['rmse']
        for k in spread:
            if k != 'rmse':
                assert c[k] <= SPREAD_FACTOR * spread[k] + 1e-12, (k, c[k], spread[k])
        assert np.array_equal(keep, g['ba2_200_keep'])
    else:
        assert res.cost <= float(g['ba2_200_cost']) * (1 + 1e-6)
        if not st['motion_reg']:
            assert rmse <= float(g['ba2_200_rmse']) + 1e-4
        assert int(np.sum(keep != g['ba2_200_keep'])) <= 3


def test_pipeline_from_files_on_disk(tmp_path):
    """the run starts from what a user of the reference has on disk -- detection text files
    (`x y frame` per line, README "2D Detection Tracks"), calibration JSON files, config.json -- goes through
    create_scene, BA -> remove_outliers -> BA on the GPU, and ends in the output pickle (main.py:93).  Checked against the
    same pipeline fed from arrays (bit for bit: the files carry %.18e) and against the reference's golden first-BA state."""
    import json
    import pickle
    scene, g = load_case('c1_pinhole_2cam')
    st = scene.settings
    dets, cams = [], []
    for i in range(scene.num_cam):
        d = scene.detections[i]
        p = tmp_path / ('cam%d.txt' % i)
        np.savetxt(p, np.column_stack((d[1], d[2], d[0])), fmt='%.18e')               # x y frame
        dets.append(str(p))
        c = scene.cameras[i]
        q = tmp_path / ('cam%d.json' % i)
        q.write_text(json.dumps({'comment': 'synthetic', 'K-matrix': c['K'].tolist(), 'distCoeff': c['d'].tolist()[:4],
                                 'fps': c['fps'], 'resolution': [int(c['resolution'][0]), int(c['resolution'][1])]}))
        cams.append(str(q))
    cfg = {'comments': 'tests/test_gpu_scene.py',
           'necessary inputs': {'path_detections': dets, 'path_cameras': cams, 'corresponding_frames': [0, 0]},
           'optional inputs': {},
           'settings': {'num_detections': 100000, 'opt_calib': False, 'cf_exact': True, 'undist_points': True,
                        'rolling_shutter': False, 'init_rs': 0, 'rs_bounds': False, 'motion_reg': False, 'motion_weights': 1,
                        'rs_bounds': False, 'camera_sequence': [0, 1], 'ref_cam': 0, 'thres_outlier': st['thres_outlier'],
                        'smooth_factor': [10, 20], 'motion_type': 'F'}}          # a reference config: no ba_* keys
    path = tmp_path / 'config.json'
    path.write_text(json.dumps(cfg))
    flight = common.create_scene(str(path))
    assert flight.numCam == 2 and flight.sequence == [0, 1] and not flight.find_order
    for i in range(2):
        np.testing.assert_array_equal(flight.detections[i], scene.detections[i])     # the text files round-trip exactly
    # what initialisation (out of scope: epipolar geometry, PnP, triangulation) would have produced -- taken from the fixture
    for i, c in enumerate(flight.cameras):
        c.R, c.t = scene.cameras[i]['R'].copy(), scene.cameras[i]['t'].copy()
        c.compose()
    flight.alpha, flight.beta = scene.alpha.copy(), scene.beta.copy()
    flight.spline = {'tck': [[t.copy(), [c.copy() for c in cs], 3] for t, cs, _ in scene.tck], 'int': scene.interval.copy()}
    flight.detection_to_global()
    ref = build_scene(scene)
    assert flight.ba_mode() == ref.ba_mode() == (_lib.SOLVER_TRF_LSMR, _lib.JAC_FD)   # the reference's algorithm, from a reference config
    before = np.array([np.mean(flight.error_cam(i)) for i in range(2)])
    np.testing.assert_allclose(before, g['mean_err_before'], rtol=0, atol=1e-9)
    r1, r1_ref = flight.BA(2), ref.BA(2)
    np.testing.assert_array_equal(r1.x, r1_ref.x)                                    # same bits from disk as from memory
    flight.remove_outliers(flight.sequence[:2], thres=flight.settings['thres_outlier'])
    ref.remove_outliers(ref.sequence[:2], thres=st['thres_outlier'])
    assert [d.shape for d in flight.detections] == [d.shape for d in ref.detections]
    r2 = flight.BA(2)
    assert r2.cost < r2.initial_cost
    out = tmp_path / 'flight.pkl'
    with open(out, 'wb') as fh:
        pickle.dump(flight, fh)
    with open(out, 'rb') as fh:
        back = pickle.load(fh)
    for name in ('alpha', 'beta', 'rs', 'sequence', 'settings', 'numCam'):          # README "Outputs"
        assert hasattr(back, name)
    np.testing.assert_array_equal(back.alpha, flight.alpha)
    np.testing.assert_array_equal(back.spline['tck'][0][1][0], flight.spline['tck'][0][1][0])
    np.testing.assert_array_equal(back.cameras[1].P, flight.cameras[1].P)
    assert back._ba_handle is None
