"""Load the golden vectors written by tests/golden/make_golden.py."""
import os
from types import SimpleNamespace

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['c1_pinhole_2cam', 'rs_F_2int_3cam', 'calib_KE_bounds_3cam', 'dist_fixed_2cam']
# BASELINE configs[4] (opt_calib + rs_bounds + KE) on a WELL-POSED scene: 5 cameras x ~3k detections, the target sweeps the
# images.  15k detections: used by the tests that need it (residual, mask, pattern, converged parity), not by every sweep.
CALIB_WP = 'calib_KE_wellposed_5cam'
# BASELINE configs[1] in shape (7 cameras, rolling shutter, motion_reg F, weight 1e4, two intervals) at 1/14 of its size
CONFIG1_SHAPE = 'config1_shape_7cam'
CONVERGED_CASES = CASES + [CALIB_WP, CONFIG1_SHAPE]


def load_case(name):
    """Return (scene, g): a scene object with the reference's attribute names and the raw npz dict."""
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False))
    C = int(g['num_cam'])
    off = g['det_offsets']
    cams = [dict(K=g['cam_K'][i], d=g['cam_d'][i], R=g['cam_R'][i], t=g['cam_t'][i],
                 fps=float(g['cam_fps'][i]), resolution=[float(g['cam_res'][i][0]), float(g['cam_res'][i][1])])
            for i in range(C)]
    dets = [g['detections'][:, off[i]:off[i + 1]].copy() for i in range(C)]
    koff = g['knot_offsets']
    tck, pos = [], 0
    for s in range(koff.size - 1):
        t = g['knots'][koff[s]:koff[s + 1]].copy()
        n = t.size - 4
        c = g['coefs'][pos:pos + 3 * n].reshape(3, n)
        pos += 3 * n
        tck.append([t, [c[0].copy(), c[1].copy(), c[2].copy()], 3])
    settings = dict(opt_calib=bool(g['opt_calib']), undist_points=bool(g['undist_points']),
                    rolling_shutter=bool(g['rolling_shutter']), rs_bounds=bool(g['rs_bounds']),
                    motion_reg=bool(g['motion_reg']), motion_type=str(g['motion_type']),
                    motion_weights=float(g['motion_weights']), thres_outlier=float(g['thres_outlier']),
                    smooth_factor=[10, 20], ref_cam=0, camera_sequence=list(range(C)))
    scene = SimpleNamespace(cameras=cams, detections=dets, alpha=g['alpha'].copy(), beta=g['beta'].copy(),
                            rs=g['rs'].copy(), tck=tck, interval=g['interval'].copy(), settings=settings,
                            num_cam=C, num_obs=int(off[-1]))
    return scene, g


def load_ensemble(name):
    """The reference's own reproducibility (tests/golden/make_golden_ensemble.py): converged parameter vectors of its
    second BA re-run with last-place noise on the residuals, two noise models x 8 members."""
    return dict(np.load(os.path.join(GOLDEN_DIR, 'ens_' + name + '.npz'), allow_pickle=False))


def reference_spread(oprob, name, x_ref):
    """Per gauge-invariant metric (tests/gauge.py) and for the final RMSE: the largest deviation of any ensemble member
    (either noise model) from the reference's unperturbed answer."""
    import gauge
    ens = load_ensemble(name)
    a = gauge.ensemble_spread(oprob, x_ref, ens['ens_x'])
    b = gauge.ensemble_spread(oprob, x_ref, ens['ensu_x'])
    return {k: max(a[k], b[k]) for k in a}


def ground_truth_x(oprob, name):
    """The generator's ground truth as a parameter vector of this problem (the fixtures do not store it; mvus_amd.synth is
    deterministic in its arguments, tests/golden_cases.py has them)."""
    import golden_cases
    from oracle import ba_oracle as orc
    tr = golden_cases.make(name).truth
    return orc.pack_x(oprob, tr['alpha'], tr['beta'], tr['rs'], tr['cameras'], [t[1] for t in tr['tck']])


def reference_spread_10(name):
    """The REAL reference's own spread of its unconverged 10-evaluation BAs under last-place noise on the residuals
    (tests/golden/make_golden_ensemble10.py): relative cost, RMSE [px] and inlier-mask flips of the first BA (`ba10`), relative cost and
    RMSE of the second (`ba2_10`)."""
    e = dict(np.load(os.path.join(GOLDEN_DIR, 'ens10_' + name + '.npz'), allow_pickle=False))
    return {'ba10': dict(cost=float(e['ba10_spread_cost_rel']), rmse=float(e['ba10_spread_rmse']), flips=int(e['ba10_spread_flips'])),
            'ba2_10': dict(cost=float(e['ba2_10_spread_cost_rel']), rmse=float(e['ba2_10_spread_rmse']))}
