#!/usr/bin/env python3
"""Golden vectors for the minimal solver of the PnP step, produced by the REAL reference's own direct-linear-transform PnP,
``epipolar.solve_PnP`` (reconstruction/epipolar.py:298-308; the hypothesis generator of its ``solve_PnP_Ransac`` :334-355, six
points per sample like this build's k_pnp_hypotheses).  The reference's ``Scene.get_camera_pose`` itself calls OpenCV
(common.py:744), which is not installed, so this is the one piece of PnP arithmetic in the reference that can be run here.
On exact correspondences every DLT formulation recovers the same projection matrix; that is what the fixture pins.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_pnp.py
Stores data only (inputs + the reference's outputs) in tests/golden/pnp_dlt.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference   # noqa: E402


def main():
    import_reference()
    from reconstruction import epipolar as ep
    rng = np.random.default_rng(77)
    K = np.array([[1100.0, 0.0, 960.0], [0.0, 1080.0, 540.0], [0.0, 0.0, 1.0]])
    out = dict(K=K)
    Xs, xs, Ps = [], [], []
    for case in range(12):
        ang = rng.normal(0, 0.5, 3)
        th = np.linalg.norm(ang)
        k = ang / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)
        t = np.array([rng.normal(0, 1), rng.normal(0, 1), 25.0 + rng.uniform(0, 10)])
        X = rng.normal(0, 4.0, (3, 6))
        Xh = np.vstack((X, np.ones(6)))
        x = K @ (R @ X + t.reshape(3, 1))
        x = x / x[2]
        P = ep.solve_PnP(x, Xh)                       # homogeneous pixels (3, 6), homogeneous points (4, 6)
        P = P / np.linalg.norm(P) * np.sign(np.linalg.det(P[:, :3]))
        Xs.append(X); xs.append(x[:2]); Ps.append(P)
    out['X'], out['x'], out['P_ref'] = np.array(Xs), np.array(xs), np.array(Ps)
    path = os.path.join(HERE, 'pnp_dlt.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d six-point samples' % (path, len(Xs)))


if __name__ == '__main__':
    main()
