#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Run in the build container only (it needs /root/reference):

    python tests/golden/make_golden.py

The reference (CenekAlbl/mvus, multiviewunsynch/reconstruction/common.py) is imported from
/root/reference -- it is never copied.  Inputs come from this repo's seeded generator
(mvus_amd/synth.py); what is stored is data only: the inputs, and the reference's outputs
(packed x0, sparsity pattern, error_BA values, least_squares results, outlier masks).

OpenCV is not installed in this image, and reconstruction/common.py imports cv2 at module
top, so a stand-in module providing the two functions the hot path calls is registered first
(SURVEY.md Appendix A): Rodrigues (via scipy.spatial.transform.Rotation) and undistortPoints
(5 fixed-point iterations, OpenCV's default TermCriteria(MAX_ITER,5,0.01)).  Vectors from
scenes with non-zero distortion are therefore "shim-defined"; pinhole ones are exact.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def install_cv2_standin():
    from scipy.spatial.transform import Rotation
    cv2 = types.ModuleType('cv2')

    def Rodrigues(src):
        a = np.asarray(src, dtype=np.float64)
        if a.size == 3:
            return Rotation.from_rotvec(a.reshape(3)).as_matrix(), None
        return Rotation.from_matrix(a.reshape(3, 3)).as_rotvec().reshape(3, 1), None

    def undistortPoints(src, K, d):
        src = np.asarray(src, dtype=np.float64).reshape(-1, 2)
        K = np.asarray(K, dtype=np.float64)
        k1, k2, p1, p2, k3 = [float(v) for v in np.asarray(d, dtype=np.float64).reshape(-1)[:5]]
        x0 = (src[:, 0] - K[0, 2]) / K[0, 0]
        y0 = (src[:, 1] - K[1, 2]) / K[1, 1]
        x, y = x0.copy(), y0.copy()
        stopped = np.zeros(x.shape, dtype=bool)      # OpenCV >= 4.1.1: a negative icdist ends the iteration with the point back at its start
        for _ in range(5):
            r2 = x * x + y * y
            with np.errstate(divide='ignore', invalid='ignore'):
                icd = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
            stopped = stopped | (icd < 0)
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x = np.where(stopped, x0, (x0 - dx) * icd)
            y = np.where(stopped, y0, (y0 - dy) * icd)
        return np.stack((x, y), axis=1).reshape(-1, 1, 2)

    cv2.Rodrigues = Rodrigues
    cv2.undistortPoints = undistortPoints
    cv2.FM_RANSAC, cv2.FM_LMEDS, cv2.FM_8POINT = 8, 4, 2
    sys.modules['cv2'] = cv2


def import_reference():
    install_cv2_standin()
    import matplotlib
    matplotlib.use('Agg')
    if not hasattr(np, 'asfarray'):
        np.asfarray = lambda a, dtype=np.float64: np.asarray(a, dtype=dtype)
    sys.path.insert(0, '/root/reference/multiviewunsynch')
    import warnings
    warnings.simplefilter('ignore')
    from reconstruction import common
    return common


def build_reference_scene(common, sc):
    """Hand-build the reference Scene from a synthetic scene."""
    ref = common.Scene()
    ref.numCam = sc.num_cam
    ref.settings = dict(sc.settings)
    for cam in sc.cameras:
        c = common.Camera(K=cam['K'].copy(), d=cam['d'].copy(), R=cam['R'].copy(), t=cam['t'].copy(),
                          fps=cam['fps'], resolution=list(cam['resolution']))
        c.compose()
        ref.addCamera(c)
    for det in sc.detections:
        ref.addDetection(det.copy())
    ref.alpha, ref.beta, ref.rs = sc.alpha.copy(), sc.beta.copy(), sc.rs.copy()
    ref.sequence = list(range(sc.num_cam))
    ref.ref_cam = 0
    ref.spline = {'tck': [[t.copy(), [c.copy() for c in cs], 3] for t, cs, _ in sc.tck],
                  'int': sc.interval.copy()}
    ref.detection_to_global()
    return ref


def snapshot_inputs(sc):
    out = {}
    C = sc.num_cam
    out['num_cam'] = np.int64(C)
    out['det_offsets'] = np.concatenate(([0], np.cumsum([d.shape[1] for d in sc.detections]))).astype(np.int64)
    out['detections'] = np.hstack(sc.detections)
    out['cam_K'] = np.array([c['K'] for c in sc.cameras])
    out['cam_d'] = np.array([c['d'] for c in sc.cameras])
    out['cam_R'] = np.array([c['R'] for c in sc.cameras])
    out['cam_t'] = np.array([c['t'] for c in sc.cameras])
    out['cam_fps'] = np.array([c['fps'] for c in sc.cameras])
    out['cam_res'] = np.array([c['resolution'] for c in sc.cameras], dtype=np.float64)
    out['alpha'], out['beta'], out['rs'] = sc.alpha.copy(), sc.beta.copy(), sc.rs.copy()
    out['knot_offsets'] = np.concatenate(([0], np.cumsum([t[0].size for t in sc.tck]))).astype(np.int64)
    out['knots'] = np.concatenate([t[0] for t in sc.tck])
    out['coefs'] = np.concatenate([np.ravel(np.asarray(t[1])) for t in sc.tck])
    out['interval'] = sc.interval.copy()
    st = sc.settings
    out['opt_calib'] = np.int64(st['opt_calib'])
    out['undist_points'] = np.int64(st['undist_points'])
    out['rolling_shutter'] = np.int64(st['rolling_shutter'])
    out['rs_bounds'] = np.int64(st['rs_bounds'])
    out['motion_reg'] = np.int64(st['motion_reg'])
    out['motion_type'] = np.array(st['motion_type'])
    out['motion_weights'] = np.float64(st['motion_weights'])
    out['thres_outlier'] = np.float64(st['thres_outlier'])
    return out


def run_case(common, name, sc, max_iters=(10,), seed=123, max_iters2=(10,)):
    from scipy import sparse
    st = sc.settings
    C = sc.num_cam
    kw = dict(rs=st['rolling_shutter'], motion_reg=st['motion_reg'], motion_weights=st['motion_weights'],
              rs_bounds=st['rs_bounds'])
    out = snapshot_inputs(sc)
    real_ls = common.least_squares

    for k, mi in enumerate(max_iters):
        ref = build_reference_scene(common, sc)
        cap = {}

        def spy(fn, x0, **kwargs):
            cap['x0'] = np.array(x0, dtype=np.float64)
            cap['A'] = kwargs['jac_sparsity']
            if k == 0:
                rng = np.random.default_rng(seed)
                delta = rng.normal(0, 1e-3, x0.size) * np.maximum(1.0, np.abs(x0)) * 1e-1
                if st['rs_bounds']:
                    pass
                cap['f0'] = np.array(fn(cap['x0']))
                cap['delta'] = delta
                cap['f1'] = np.array(fn(cap['x0'] + delta))
                fn(cap['x0'])            # put the scene state back to x0
            return real_ls(fn, x0, **kwargs)

        common.least_squares = spy
        try:
            err_before = np.array([np.mean(ref.error_cam(i)) for i in range(C)])
            res = ref.BA(C, max_iter=mi, **kw)
            err_after = np.array([np.mean(ref.error_cam(i)) for i in range(C)])
        finally:
            common.least_squares = real_ls

        if k == 0:
            A = sparse.coo_matrix(np.asarray(cap['A']))
            out['x0'] = cap['x0']
            out['pattern_shape'] = np.array(A.shape, dtype=np.int64)
            order = np.lexsort((A.col, A.row))
            out['pattern_rows'] = A.row[order].astype(np.int32)
            out['pattern_cols'] = A.col[order].astype(np.int32)
            out['f_x0'] = cap['f0']
            out['delta'] = cap['delta']
            out['f_x0_delta'] = cap['f1']
            out['mean_err_before'] = err_before
        tag = 'ba%d' % mi
        out[tag + '_x'] = np.array(res.x)
        out[tag + '_cost'] = np.float64(res.cost)
        out[tag + '_fun'] = np.array(res.fun)
        out[tag + '_nfev'] = np.int64(res.nfev)
        out[tag + '_njev'] = np.int64(res.njev)
        out[tag + '_status'] = np.int64(res.status)
        out[tag + '_optimality'] = np.float64(res.optimality)
        out[tag + '_mean_err_after'] = err_after
        rmse = np.sqrt(np.mean(np.concatenate([ref.error_cam(i, 'dist') for i in range(C)]) ** 2))
        out[tag + '_rmse'] = np.float64(rmse)
        print('  %s max_iter=%d: n=%d m=%d cost=%.9g nfev=%d njev=%d status=%d rmse=%.6f'
              % (name, mi, res.x.size, res.fun.size, res.cost, res.nfev, res.njev, res.status, rmse))

        if k == 0:
            # outlier pass exactly as main.py:56 after the first BA
            if st['motion_reg']:
                # Scene.all_detect_to_traj at the state BA leaves behind (common.py:887-947): the attributes of the output pickle
                ref.all_detect_to_traj(ref.sequence[:C])
                out['adt_global_traj'] = np.array(ref.global_traj)
                out['adt_global_detections'] = np.array(ref.global_detections)
                out['adt_frame_id_all'] = np.array(ref.frame_id_all)
                out['adt_global_time_stamps_all'] = np.array(ref.global_time_stamps_all)
                out['adt_traj'] = np.array(ref.traj)
            frames_before = [d[0].copy() for d in ref.detections]
            ref.remove_outliers(ref.sequence[:C], thres=st['thres_outlier'])
            keep = [np.isin(fb, d[0]) for fb, d in zip(frames_before, ref.detections)]
            out['outlier_keep'] = np.concatenate(keep).astype(np.uint8)
            print('  %s outliers removed: %d of %d' % (name, int((~np.concatenate(keep)).sum()), out['outlier_keep'].size))
            # second BA on the filtered detections, main.py:59 (the run's final answer)
            for mi2 in max_iters2:
                ref2 = build_reference_scene(common, sc)
                res1 = ref2.BA(C, max_iter=mi, **kw)
                ref2.remove_outliers(ref2.sequence[:C], thres=st['thres_outlier'])
                cap2 = {}

                def spy2(fn, x0, **kwargs):
                    cap2['x0'] = np.array(x0, dtype=np.float64)
                    cap2['A'] = kwargs['jac_sparsity']
                    return real_ls(fn, x0, **kwargs)
                common.least_squares = spy2
                try:
                    res2 = ref2.BA(C, max_iter=mi2, **kw)
                finally:
                    common.least_squares = real_ls
                t2 = 'ba2_%d' % mi2
                out[t2 + '_x0'] = cap2['x0']
                A2 = sparse.coo_matrix(np.asarray(cap2['A']))          # the matrix of the second BA (same x0 for every mi2)
                order2 = np.lexsort((A2.col, A2.row))
                out['ba2_pattern_shape'] = np.array(A2.shape, dtype=np.int64)
                out['ba2_pattern_rows'] = A2.row[order2].astype(np.int32)
                out['ba2_pattern_cols'] = A2.col[order2].astype(np.int32)
                out[t2 + '_x'] = np.array(res2.x)
                out[t2 + '_cost'] = np.float64(res2.cost)
                out[t2 + '_nfev'] = np.int64(res2.nfev)
                out[t2 + '_njev'] = np.int64(res2.njev)
                out[t2 + '_status'] = np.int64(res2.status)
                out[t2 + '_optimality'] = np.float64(res2.optimality)
                rmse2 = np.sqrt(np.mean(np.concatenate([ref2.error_cam(i, 'dist') for i in range(C)]) ** 2))
                out[t2 + '_rmse'] = np.float64(rmse2)
                # the reference's own inlier mask at this (converged) point: remove_outliers once more, on a copy
                # (over the detections the second BA ran on, i.e. those kept by 'outlier_keep')
                frames2 = [d[0].copy() for d in ref2.detections]
                ref2.remove_outliers(ref2.sequence[:C], thres=st['thres_outlier'])
                out[t2 + '_keep'] = np.concatenate([np.isin(fb, d[0]) for fb, d in zip(frames2, ref2.detections)]).astype(np.uint8)
                print('  %s second BA max_iter=%d: cost=%.9g nfev=%d njev=%d status=%d rmse=%.6f'
                      % (name, mi2, res2.cost, res2.nfev, res2.njev, res2.status, rmse2))

    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote %s (%.1f KiB)' % (path, os.path.getsize(path) / 1024))


def main():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import golden_cases
    common = import_reference()
    only = sys.argv[1:]
    for name in golden_cases.GENERATORS:
        if only and name not in only:
            continue
        sc = golden_cases.make(name)
        print('case %s: C=%d M=%d' % (name, sc.num_cam, sc.num_obs))
        run_case(common, name, sc, golden_cases.MAX_ITERS.get(name, (10,)), max_iters2=(10, 200))


if __name__ == '__main__':
    main()
