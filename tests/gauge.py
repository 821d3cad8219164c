"""Gauge-invariant comparison of two BA solutions of the same problem (test infrastructure).

What the caller of Scene.BA reads back is res.x only (reference common.py:672-695): alpha, beta, rs, the camera
poses (and K, d with opt_calib) and the spline control points.  The objective does not see a joint similarity
transform of cameras + trajectory (7 degrees of freedom; every camera is free in the reference's BA), and it is almost
blind to a joint shift of all beta with the curve re-timed accordingly.  Two runs that agree on every observable can
therefore differ in x by far more than they differ in anything measurable.  `compare` reports:

  traj_rms / traj_max   3-D distance (scene units, metres in the synthetic scenes) between the two trajectories evaluated
                        at every detection's own time stamp (alpha (f + rs v / H) + beta of the respective solution),
                        after the best similarity (Umeyama, with scale) of solution b onto solution a
  centre_max            distance between corresponding camera centres under that same similarity
  rot_max_deg           largest angle between corresponding camera orientations under it
  dbeta_max             max_c |(beta_c - beta_0)_a - (beta_c - beta_0)_b|   (frames of the reference camera)
  alpha_max             max_c |(alpha_c / alpha_0)_a - (alpha_c / alpha_0)_b|
  rs_max                max_c |rs_a - rs_b|
  K_rel_max, d_max      (opt_calib) max relative difference of fx, fy, cx, cy; max absolute difference of the five
                        distortion coefficients
  scale, rmse_a, rmse_b the similarity's scale; reprojection RMSE of both (px)
"""
import numpy as np

from oracle import ba_oracle as orc


def umeyama(src, dst):
    """Similarity (s, R, t) minimising sum |dst - (s R src + t)|^2 (Umeyama 1991); src, dst: 3 x K."""
    mu_s, mu_d = src.mean(axis=1, keepdims=True), dst.mean(axis=1, keepdims=True)
    a, b = src - mu_s, dst - mu_d
    U, S, Vt = np.linalg.svd(b @ a.T / src.shape[1])
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1.0
    R = U @ D @ Vt
    s = float(np.trace(np.diag(S) @ D) / (a * a).sum() * src.shape[1])
    t = mu_d - s * R @ mu_s
    return s, R, t


def trajectory_at_detections(oprob, x):
    """Per camera: (inside mask, X[3, M_c]) -- the spline point at every detection's time stamp under solution x."""
    alpha, beta, rs, cams, tck = orc.unpack_x(oprob, np.asarray(x, dtype=np.float64))
    out = []
    for c in range(oprob.C):
        tau = orc.detection_to_global(oprob, c, alpha, beta, rs, cams[c])[0]
        idx = orc.sampling_idx(tau, oprob.interval)
        X = np.zeros((3, tau.size))
        for s in range(oprob.interval.shape[1]):
            m = idx == s + 1
            if m.any():
                X[:, m] = np.asarray(orc.splev3(tau[m], tck[s]))
        out.append((idx > 0, X))
    return out


def compare(oprob, xa, xb):
    xa, xb = np.asarray(xa, dtype=np.float64), np.asarray(xb, dtype=np.float64)
    C, P = oprob.C, oprob.P
    ta, tb = trajectory_at_detections(oprob, xa), trajectory_at_detections(oprob, xb)
    both = [ma & mb for (ma, _), (mb, _) in zip(ta, tb)]
    Xa = np.hstack([X[:, m] for (_, X), m in zip(ta, both)])
    Xb = np.hstack([X[:, m] for (_, X), m in zip(tb, both)])
    s, R, t = umeyama(Xb, Xa)
    d = np.sqrt(((Xa - (s * R @ Xb + t)) ** 2).sum(axis=0))
    _, _, _, cams_a, _ = orc.unpack_x(oprob, xa)
    _, _, _, cams_b, _ = orc.unpack_x(oprob, xb)
    centre, rot = [], []
    for ca, cb in zip(cams_a, cams_b):
        ca_c = -ca['R'].T @ ca['t']
        cb_c = s * R @ (-cb['R'].T @ cb['t']) + t[:, 0]
        centre.append(np.linalg.norm(ca_c - cb_c))
        Rrel = ca['R'] @ (cb['R'] @ R.T).T               # camera b's orientation expressed in a's world frame
        rot.append(np.degrees(np.arccos(np.clip(0.5 * (np.trace(Rrel) - 1.0), -1.0, 1.0))))
    al_a, be_a, rs_a = xa[:C], xa[C:2 * C], xa[2 * C:3 * C]
    al_b, be_b, rs_b = xb[:C], xb[C:2 * C], xb[2 * C:3 * C]
    out = dict(traj_rms=float(np.sqrt(np.mean(d ** 2))), traj_max=float(d.max()), centre_max=float(max(centre)),
               rot_max_deg=float(max(rot)), scale=float(s),
               dbeta_max=float(np.max(np.abs((be_a - be_a[0]) - (be_b - be_b[0])))),
               alpha_max=float(np.max(np.abs(al_a / al_a[0] - al_b / al_b[0]))),
               rs_max=float(np.max(np.abs(rs_a - rs_b))),
               rmse_a=orc.reprojection_rmse(oprob, xa), rmse_b=orc.reprojection_rmse(oprob, xb),
               n_points=int(d.size))
    if oprob.opt_calib:
        ka = np.array([xa[3 * C + i * P: 3 * C + i * P + 4] for i in range(C)])
        kb = np.array([xb[3 * C + i * P: 3 * C + i * P + 4] for i in range(C)])
        da = np.array([xa[3 * C + i * P + 10: 3 * C + (i + 1) * P] for i in range(C)])
        db = np.array([xb[3 * C + i * P + 10: 3 * C + (i + 1) * P] for i in range(C)])
        out['K_rel_max'] = float(np.max(np.abs(ka - kb) / np.abs(ka)))
        out['d_max'] = float(np.max(np.abs(da - db)))
    return out


METRICS = ('traj_rms', 'traj_max', 'centre_max', 'rot_max_deg', 'dbeta_max', 'alpha_max', 'rs_max')
CALIB_METRICS = ('K_rel_max', 'd_max')


def ensemble_spread(oprob, x_ref, ens_x):
    """Largest value of every metric over the members of the reference's own ensemble (each compared with x_ref), plus the
    spread of the final RMSE: what the reference itself reproduces under 1e-15 relative noise on its residuals."""
    rows = [compare(oprob, x_ref, xe) for xe in ens_x]
    keys = METRICS + (CALIB_METRICS if oprob.opt_calib else ())
    spread = {k: max(r[k] for r in rows) for k in keys}
    spread['rmse'] = max(abs(r['rmse_b'] - r['rmse_a']) for r in rows)
    return spread
