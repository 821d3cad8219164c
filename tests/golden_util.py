"""Load the golden vectors written by tests/golden/make_golden.py."""
import os
from types import SimpleNamespace

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['c1_pinhole_2cam', 'rs_F_2int_3cam', 'calib_KE_bounds_3cam', 'dist_fixed_2cam']


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
