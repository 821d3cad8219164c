"""The incremental reconstruction loop of the reference's ``main.py`` (lines 43-82) over the drop-in ``Scene``: for the
current number of cameras BA -> remove_outliers -> BA, then -- until every camera is in -- select_most_overlap ->
get_camera_pose -> triangulate (with the spline refit).  Every stage runs on the GPU through libmvusba.so; this module only
sequences them (same order, same settings keys) and keeps a wall-clock split per stage.

Out of scope stays out of scope: the start of the loop -- two cameras with poses and a first trajectory, which the reference
gets from ``init_traj`` (epipolar geometry) -- is an input here.
"""
import time

import numpy as np


class StageTimer:
    """Wall-clock per stage, in call order (the reference prints only a running total, main.py:68,81)."""

    def __init__(self):
        self.rows = []

    def run(self, stage, num_cam, fn, *args, **kw):
        t0 = time.perf_counter()
        out = fn(*args, **kw)
        self.rows.append((stage, num_cam, time.perf_counter() - t0))
        return out

    def totals(self):
        tot = {}
        for stage, _, dt in self.rows:
            tot[stage] = tot.get(stage, 0.0) + dt
        return tot


def incremental_reconstruction(flight, max_iter=10, verbose=False, timer=None):
    """main.py:43-82.  ``flight``: a ``Scene`` whose first two cameras of ``sequence`` (or of ``select_most_overlap``) have
    poses and whose ``spline`` holds the first trajectory; the remaining cameras have ``P is None``.  Uses the settings keys
    of the reference's config.json: rolling_shutter, motion_reg, motion_weights, rs_bounds, thres_outlier,
    thres_triangulation, smooth_factor, sampling_rate, thres_PnP (optional, default 8 as in get_camera_pose).  With the opt-in
    ``ba_solver: 'lm'`` and no ``ba_lambda_min`` the loop sets its own damping floor (0.3) in ``flight.settings``.
    Returns the StageTimer."""
    timer = timer or StageTimer()
    st = flight.settings
    loop_floor = st.get('ba_solver') == 'lm' and 'ba_lambda_min' not in st
    if loop_floor:
        # the staged, gauge-free BAs of the loop need more damping than one large BA (reconstruction/common.py: LOOP_LM_LAMBDA_MIN);
        # the floor is in force for the loop only: a later stand-alone Scene.BA on this Scene gets the library's own again
        from .reconstruction.common import LOOP_LM_LAMBDA_MIN
        st['ba_lambda_min'] = LOOP_LM_LAMBDA_MIN
        print("incremental_reconstruction: ba_solver = 'lm' without ba_lambda_min -- using the loop's damping floor %g" % LOOP_LM_LAMBDA_MIN)
    try:
        return _incremental_loop(flight, st, max_iter, verbose, timer)
    finally:
        if loop_floor:
            st.pop('ba_lambda_min', None)


def _incremental_loop(flight, st, max_iter, verbose, timer):
    kw = dict(rs=st['rolling_shutter'], motion_reg=st['motion_reg'], motion_weights=st['motion_weights'], rs_bounds=st['rs_bounds'])
    cam_temp = 2
    while True:
        seq = flight.sequence[:cam_temp]
        if verbose:
            print('\n----------------- Bundle Adjustment with {} cameras -----------------'.format(cam_temp))
            print('Mean error of each camera before BA:   ', np.asarray([np.mean(flight.error_cam(x)) for x in seq]))
        timer.run('BA', cam_temp, flight.BA, cam_temp, max_iter=max_iter, **kw)
        timer.run('remove_outliers', cam_temp, flight.remove_outliers, seq, thres=st['thres_outlier'])
        timer.run('BA', cam_temp, flight.BA, cam_temp, max_iter=max_iter, **kw)
        if verbose:
            print('Mean error of each camera after second BA:    ', np.asarray([np.mean(flight.error_cam(x)) for x in seq]))
        num_end = flight.numCam if flight.find_order else len(flight.sequence)
        if cam_temp == num_end:
            break
        timer.run('select_most_overlap', cam_temp, flight.select_most_overlap)
        nxt = flight.sequence[cam_temp]
        timer.run('get_camera_pose', cam_temp, flight.get_camera_pose, nxt, error=st.get('thres_PnP', 8))
        timer.run('triangulate', cam_temp, flight.triangulate, nxt, flight.sequence[:cam_temp], thres=st['thres_triangulation'],
                  factor_t2s=st['smooth_factor'], factor_s2t=st['sampling_rate'])
        cam_temp += 1
        flight.traj_len = []
    timer.run('spline_to_traj', cam_temp, flight.spline_to_traj, sampling_rate=1)
    return timer


def staged_scene(num_cam=7, total_obs=100_000, seed=2, windows=None, settings=None, **scene_kw):
    """A synthetic scene in the state the reference's loop starts from (after ``init_traj``): cameras observe the target during
    different parts of the flight, the first two have (perturbed) poses and the trajectory exists over THEIR common time range
    only; the others have no pose yet.  ``windows[i]`` = (from, to) as fractions of the flight for camera i.  Returns
    (Scene, SynthScene) -- the latter carries the generator's ground truth."""
    from . import synth
    from .reconstruction import common
    sc = synth.make_scene(num_cam, total_obs, seed=seed, **scene_kw)
    if windows is None:
        windows = [(0.0, 0.8), (0.0, 0.55), (0.1, 0.75), (0.3, 1.0), (0.2, 1.0)] + [(0.0, 1.0)] * max(num_cam - 5, 0)
    G = float(sc.interval[1, -1])
    tr = sc.truth
    for i in range(num_cam):
        d = sc.detections[i]
        tau = tr['alpha'][i] * d[0] + tr['beta'][i]
        lo, hi = windows[i][0] * G, windows[i][1] * G
        sc.detections[i] = d[:, (tau >= lo) & (tau <= hi)]
    s = common.Scene()
    s.numCam = num_cam
    s.settings = dict(sc.settings)
    s.settings.update(sampling_rate=0.02, thres_triangulation=20, thres_PnP=8)
    s.settings.update(settings or {})
    for c in sc.cameras:
        cam = common.Camera(K=c['K'].copy(), d=c['d'].copy(), R=c['R'].copy(), t=c['t'].copy(), fps=c['fps'], resolution=list(c['resolution']))
        cam.compose()
        s.addCamera(cam)
    for det in sc.detections:
        s.addDetection(det.copy())
    s.alpha, s.beta, s.rs = sc.alpha.copy(), sc.beta.copy(), sc.rs.copy()
    s.sequence, s.find_order, s.ref_cam = [0, 1], True, 0
    for i in range(2, num_cam):
        s.cameras[i].R = s.cameras[i].t = s.cameras[i].P = None          # poses to be found by get_camera_pose
    # the first trajectory: the generator's (perturbed) spline sampled over the common range of cameras 0 and 1, refitted
    s.spline = {'tck': [[t.copy(), [c.copy() for c in cs], 3] for t, cs, _ in sc.tck], 'int': sc.interval.copy()}
    s.detection_to_global()
    lo = max(s.detections_global[0][0].min(), s.detections_global[1][0].min())
    hi = min(s.detections_global[0][0].max(), s.detections_global[1][0].max())
    s.spline_to_traj(sampling_rate=1)
    s.traj = s.traj[:, (s.traj[0] >= lo) & (s.traj[0] <= hi)]
    s.traj_to_spline(smooth_factor=s.settings['smooth_factor'])
    s.detection_to_global([0, 1])
    return s, sc


def evaluate_against_truth(flight, sc):
    """Accuracy of a finished reconstruction against the generator's ground truth (``sc.truth``): per-camera mean reprojection
    error of the detections that were kept; how the kept set relates to the generator's clean detections (within 5 px = 10
    sigma of the true projection); trajectory, camera centres and orientations after the best similarity of the
    reconstructed trajectory onto the true one."""
    from . import bspline
    from .analysis.compare_gt import similarity_from_points
    tr = sc.truth
    C = flight.numCam

    def true_curve(tau):
        X = np.zeros((3, tau.size))
        inside = np.zeros(tau.size, dtype=bool)
        for tck in tr['tck']:
            m = (tau >= tck[0][0]) & (tau < tck[0][-1])
            if m.any():
                X[:, m] = bspline.evaluate(tck[0], np.array(tck[1]), tau[m])
            inside |= m
        return X, inside
    out = {'mean_err': [], 'kept': [], 'clean': [], 'kept_dirty': []}
    for i in range(C):
        d0 = sc.detections[i]
        tau = tr['alpha'][i] * (d0[0] + tr['rs'][i] * d0[2] / sc.cameras[i]['resolution'][1]) + tr['beta'][i]
        X, inside = true_curve(tau)
        cam = tr['cameras'][i]
        Xc = cam['R'] @ X + cam['t'][:, None]
        u = cam['K'][0, 0] * Xc[0] / Xc[2] + cam['K'][0, 2]
        v = cam['K'][1, 1] * Xc[1] / Xc[2] + cam['K'][1, 2]
        clean = inside & (np.hypot(u - d0[1], v - d0[2]) < 5.0)
        kept = np.isin(d0[0], flight.detections[i][0])
        err = flight.error_cam(i, mode='each')
        m_ = err.size // 2
        dist = np.hypot(err[:m_], err[m_:])
        out['mean_err'].append(float(np.mean(dist[dist > 0])))
        out['kept'].append(int(kept.sum()))
        out['clean'].append(int(clean.sum()))
        out['kept_dirty'].append(int((kept & ~clean & inside).sum()))
    ts = np.arange(np.ceil(flight.spline['int'][0, 0]), np.floor(flight.spline['int'][1, -1]), 1.0)
    traj = flight.spline_to_traj(t=ts)
    Xt, ok = true_curve(traj[0])
    M = similarity_from_points(traj[1:, ok], Xt[:, ok])
    sR, t = M[:3, :3], M[:3, 3]
    scale = float(np.cbrt(np.linalg.det(sR)))
    R = sR / scale
    d = np.sqrt(((Xt[:, ok] - (sR @ traj[1:, ok] + t[:, None])) ** 2).sum(axis=0))
    out['traj_rms'], out['traj_max'], out['scale'] = float(np.sqrt(np.mean(d ** 2))), float(d.max()), scale
    out['centre_err'], out['rot_err_deg'] = [], []
    for i in range(C):
        c_est = sR @ (-flight.cameras[i].R.T @ flight.cameras[i].t) + t
        c_true = -tr['cameras'][i]['R'].T @ tr['cameras'][i]['t']
        out['centre_err'].append(float(np.linalg.norm(c_est - c_true)))
        Rrel = tr['cameras'][i]['R'] @ (flight.cameras[i].R @ R.T).T
        out['rot_err_deg'].append(float(np.degrees(np.arccos(np.clip(0.5 * (np.trace(Rrel) - 1.0), -1.0, 1.0)))))
    out['trajectory_extent'] = (float(flight.spline['int'][0, 0]), float(flight.spline['int'][1, -1]), float(sc.interval[1, -1]))
    return out
