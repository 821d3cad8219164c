#!/usr/bin/env python3
"""Golden vector for the ground-truth alignment report: runs the REAL reference
``analysis/compare_gt.py:align_gt`` (imported from /root/reference, never copied) on a seeded synthetic flight and a
synthetic ground-truth track, and stores inputs + the reference's outputs in tests/golden/align_gt_2cam.npz.

    python tests/golden/make_golden_align.py        (build container only)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                      # noqa: E402  (cv2 stand-in + reference import helpers)


def main():
    common = mg.import_reference()
    from analysis import compare_gt                           # the reference's module
    from scipy.spatial.transform import Rotation
    from mvus_amd import synth
    sc = synth.baseline_scene(0)
    ref = mg.build_reference_scene(common, sc)
    ref.settings['ref_cam'] = 0
    f_gt = 5.0
    fps = ref.cameras[0].fps
    alpha_true, beta_true = fps / f_gt * 1.0003, float(sc.interval[0, 0]) + 7.3
    k = np.arange(int((sc.interval[1, -1] - beta_true) / alpha_true) - 2)
    t_gt = alpha_true * k + beta_true
    X = ref.spline_to_traj(t=t_gt)[1:]
    rng = np.random.default_rng(77)
    R = Rotation.from_rotvec([0.2, -0.4, 1.1]).as_matrix()
    gt = 0.37 * R @ X + np.array([[12.0], [-3.0], [40.0]]) + rng.normal(scale=0.01, size=X.shape)
    gt_path = os.path.join(HERE, '_gt_tmp.txt')
    np.savetxt(gt_path, gt)
    out = compare_gt.align_gt(ref, f_gt, gt_path, visualize=False)
    os.remove(gt_path)
    snap = mg.snapshot_inputs(sc)
    snap.update(dict(f_gt=np.float64(f_gt), gt=gt, align_param=np.asarray(out['align_param']), tran_matrix=out['tran_matrix'],
                     error=out['error'], error_mean=np.float64(np.mean(out['error'])), error_median=np.float64(np.median(out['error'])),
                     alpha_true=np.float64(alpha_true), beta_true=np.float64(beta_true)))
    np.savez_compressed(os.path.join(HERE, 'align_gt_2cam.npz'), **snap)
    print('align_param', out['align_param'], 'true', alpha_true, beta_true, 'mean error', np.mean(out['error']), 'n', out['error'].size)


if __name__ == '__main__':
    main()
