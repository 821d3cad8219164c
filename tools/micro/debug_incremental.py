"""Stage-by-stage diagnostics of the incremental loop (tools/incremental_loop.py) against the generator's ground truth."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from mvus_amd import pipeline, synth, bspline

def angle(Ra, Rb):
    return float(np.degrees(np.arccos(np.clip(0.5 * (np.trace(Ra @ Rb.T) - 1.0), -1.0, 1.0))))

def main():
    nobs = int(sys.argv[1]) if len(sys.argv) > 1 else 21000
    solver = sys.argv[2] if len(sys.argv) > 2 else 'trf'
    max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    over = dict(a.split('=') for a in sys.argv[4:])
    kw = dict(synth.BASELINE_CONFIGS[1]); kw.pop('seed'); kw.pop('num_cam'); kw.pop('total_obs'); kw.pop('num_intervals', None)
    for k, v in over.items():
        kw[k] = type(kw.get(k, 0.0))(eval(v)) if k in kw else eval(v)
    flight, sc = pipeline.staged_scene(7, nobs, seed=2, settings={'ba_solver': solver}, perturb=0.3, **kw)
    tr = sc.truth
    st = flight.settings
    bakw = dict(rs=st['rolling_shutter'], motion_reg=st['motion_reg'], motion_weights=st['motion_weights'], rs_bounds=st['rs_bounds'])
    def errs(cams):
        out = []
        for i in cams:
            e = flight.error_cam(i, mode='each'); m = e.size // 2; d = np.hypot(e[:m], e[m:]); d = d[d > 0]
            out.append('%d:%.2f/%.1f(n%d)' % (i, np.median(d) if d.size else -1, np.mean(d) if d.size else -1, flight.detections[i].shape[1]))
        return ' '.join(out)
    def truth_curve(t):
        X = np.zeros((3, t.size))
        for tck in tr['tck']:
            m = (t >= tck[0][0]) & (t <= tck[0][-1])
            X[:, m] = bspline.evaluate(tck[0], np.array(tck[1]), t[m])
        return X
    def traj_err():
        ts = np.arange(np.ceil(flight.spline['int'][0, 0]), np.floor(flight.spline['int'][1, -1]), 2.0)
        tj = flight.spline_to_traj(t=ts)
        d = np.linalg.norm(tj[1:] - truth_curve(tj[0]), axis=0)
        return 'traj vs truth (no alignment): rms %.3f max %.3f over %s, %d ctrl' % (np.sqrt(np.mean(d ** 2)), d.max(), flight.spline['int'].round(0).tolist(), sum(len(t[0]) - 4 for t in flight.spline['tck']))
    cam_temp = 2
    print('start', errs(flight.sequence[:2]), traj_err())
    while True:
        seq = flight.sequence[:cam_temp]
        r = flight.BA(cam_temp, max_iter=max_iter, **bakw)
        print('BA1[%d] cost %.4g -> %.4g nfev %d st %d |' % (cam_temp, r.initial_cost, r.cost, r.nfev, r.status), errs(seq), '|', traj_err())
        flight.remove_outliers(seq, thres=st['thres_outlier'])
        r = flight.BA(cam_temp, max_iter=max_iter, **bakw)
        print('BA2[%d] cost %.4g -> %.4g nfev %d st %d |' % (cam_temp, r.initial_cost, r.cost, r.nfev, r.status), errs(seq), '|', traj_err())
        print('   beta err', np.round(flight.beta[seq] - tr['beta'][seq], 3), 'rs err', np.round(flight.rs[seq] - tr['rs'][seq], 3),
              'rot err', [round(angle(flight.cameras[i].R, tr['cameras'][i]['R']), 3) for i in seq])
        if cam_temp == flight.numCam:
            break
        flight.select_most_overlap()
        nxt = flight.sequence[cam_temp]
        flight.get_camera_pose(nxt, error=8)
        print('PnP cam %d: rot err %.3f deg, centre err %.3f m' % (nxt, angle(flight.cameras[nxt].R, tr['cameras'][nxt]['R']),
              np.linalg.norm(-flight.cameras[nxt].R.T @ flight.cameras[nxt].t + tr['cameras'][nxt]['R'].T @ tr['cameras'][nxt]['t'])), errs([nxt]))
        Xn = flight.triangulate(nxt, flight.sequence[:cam_temp], thres=st['thres_triangulation'], factor_t2s=st['smooth_factor'], factor_s2t=st['sampling_rate'])
        if Xn.shape[1]:
            d = np.linalg.norm(Xn[1:] - truth_curve(Xn[0]), axis=0)
            print('triangulated %d new points, vs truth: median %.3f max %.3f m;' % (Xn.shape[1], np.median(d), d.max()), traj_err())
        else:
            print('triangulated nothing;', traj_err())
        cam_temp += 1
main()
