#!/usr/bin/env python3
"""Golden vectors for the triangulation step, produced by the REAL reference:

  * ``epipolar.triangulate_matlab`` (reconstruction/epipolar.py:497-510) and the two reprojection errors
    (``Camera.projectPoint`` + ``epipolar.reprojection_error``) on matched detections of two synthetic cameras;
  * ``Scene.triangulate`` (common.py:754-815) end to end: a spline fitted to the first 60 % of the timeline, the third
    camera's remaining detections triangulated against cameras 0 and 1 and appended, the spline refitted.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_triangulate.py
Stores data only (inputs of this repo's seeded generator + the reference's outputs) in tests/golden/triangulate_3cam.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import ROOT, build_reference_scene, import_reference, snapshot_inputs   # noqa: E402


def main():
    from mvus_amd import synth
    common = import_reference()
    from reconstruction import epipolar as ep
    from tools import util
    sc = synth.make_scene(3, 2400, seed=41, knot_spacing=15.0)
    st = sc.settings
    ref = build_reference_scene(common, sc)
    ref.settings['smooth_factor'] = [10, 20]
    out = snapshot_inputs(sc)
    # function level: matched detections of cameras 2 and 0 over the whole timeline
    x1, x2 = util.match_overlap(ref.detections_global[2], ref.detections_global[0])
    P1, P2 = ref.cameras[2].P, ref.cameras[0].P
    X = ep.triangulate_matlab(x1[1:], x2[1:], P1, P2)
    out['fn_x1'], out['fn_x2'], out['fn_P1'], out['fn_P2'], out['fn_X'] = x1, x2, P1, P2, X
    out['fn_err1'] = ep.reprojection_error(x1[1:], ref.cameras[2].projectPoint(X[:-1]))
    out['fn_err2'] = ep.reprojection_error(x2[1:], ref.cameras[0].projectPoint(X[:-1]))
    # scene level
    ref.spline_to_traj()
    t_cut = ref.traj[0, 0] + 0.6 * (ref.traj[0, -1] - ref.traj[0, 0])
    ref.traj = ref.traj[:, ref.traj[0] < t_cut]
    out['sc_traj_in'] = ref.traj.copy()
    ref.traj_to_spline(smooth_factor=[10, 20])
    out['sc_int_before'] = ref.spline['int'].copy()
    X_new = ref.triangulate(2, [0, 1], factor_t2s=[10, 20], factor_s2t=0.02, thres=20)
    out['sc_thres'] = np.float64(20)
    out['sc_X_new'] = X_new
    out['sc_traj_out_shape'] = np.array(ref.traj.shape, dtype=np.int64)      # 23k samples at 0.02: keep every 40th + the sums
    out['sc_traj_out_sub'] = ref.traj[:, ::40].copy()
    out['sc_traj_out_sum'] = ref.traj.sum(axis=1)
    out['sc_int_after'] = ref.spline['int'].copy()
    out['sc_knot_offsets_after'] = np.concatenate(([0], np.cumsum([t[0].size for t in ref.spline['tck']]))).astype(np.int64)
    out['sc_knots_after'] = np.concatenate([t[0] for t in ref.spline['tck']])
    out['sc_coefs_after'] = np.concatenate([np.ravel(np.asarray(t[1])) for t in ref.spline['tck']])
    print('function level: %d pairs; scene level: %d new points, intervals %s -> %s, %d knots'
          % (X.shape[1], X_new.shape[1], out['sc_int_before'].tolist(), out['sc_int_after'].tolist(), out['sc_knots_after'].size))
    path = os.path.join(HERE, 'triangulate_3cam.npz')
    np.savez_compressed(path, **out)
    print('wrote %s (%.1f KiB)' % (path, os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
