"""CPU ORACLE for the bundle-adjustment hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement of the reference algorithm
(CenekAlbl/mvus, ``multiviewunsynch/reconstruction/common.py`` and
``multiviewunsynch/tools/util.py``; every function cites the lines it follows).
It exists to *check* the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``mvus_amd/`` imports it
and the product path never falls back to it.

Parity status: PINNED against the reference itself.  ``tests/golden/make_golden.py`` imports
the real reference from ``/root/reference`` (with the ``cv2`` stand-in described below) and
stores its outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every
function here against those vectors.  The reference has no tests or golden vectors of its
own.

Third-party arithmetic that is *not* under ``/root/reference``:

* ``scipy`` (unpinned by the reference; 1.15.3 in this image): ``least_squares`` (call site
  ``common.py:670``) is called here exactly like the reference does -- the optimiser is not
  restated in the oracle; ``splev`` (``common.py:294,331``) is restated (``splev3``) following
  FITPACK's ``splev.f``/``fpbspl.f`` and additionally pinned against ``scipy.interpolate.splev``.
* OpenCV (unpinned; ABSENT from this image): ``cv2.Rodrigues`` (``common.py:1119,1136,1140``)
  and ``cv2.undistortPoints`` (``common.py:1154``) are restated from OpenCV's published
  algorithm (Rodrigues formula; 5 fixed-point iterations of the 5-coefficient model =
  ``TermCriteria(MAX_ITER, 5, 0.01)`` default of ``undistortPoints``).  The golden vectors were
  produced with the same stand-in, so results with non-zero distortion are "shim-defined";
  pinhole (d = 0) vectors do not depend on it.
"""
from dataclasses import dataclass, field
import numpy as np


# ----------------------------------------------------------------------------------------
# OpenCV stand-ins
# ----------------------------------------------------------------------------------------
def rodrigues(rvec):
    """cv2.Rodrigues, vector -> matrix (call sites common.py:1136,1140)."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / theta
    c, s = np.cos(theta), np.sin(theta)
    Kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return c * np.eye(3) + (1.0 - c) * np.outer(k, k) + s * Kx


def rotation_to_rvec(R):
    """cv2.Rodrigues, matrix -> vector (call site common.py:1119)."""
    R = np.asarray(R, dtype=np.float64)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = 0.5 * np.sqrt(w @ w)
    c = np.clip(0.5 * (np.trace(R) - 1.0), -1.0, 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        B = 0.5 * (R + np.eye(3))
        i = int(np.argmax(np.diag(B)))
        ax = B[:, i] / np.sqrt(B[i, i])
        return ax * (theta / np.sqrt(ax @ ax))
    return w * (0.5 * theta / s)


def undistort_normalized(uv, K, d, iters=5):
    """cv2.undistortPoints(src, K, d): normalised coordinates, 5 fixed-point iterations."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    k1, k2, p1, p2, k3 = [float(v) for v in np.asarray(d, dtype=np.float64).reshape(-1)[:5]]
    x0 = (uv[0] - cx) / fx
    y0 = (uv[1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    stopped = np.zeros(np.shape(x), dtype=bool)       # OpenCV >= 4.1.1: icdist < 0 ends the iteration with the point back at (x0, y0)
    for _ in range(iters):
        r2 = x * x + y * y
        with np.errstate(divide='ignore', invalid='ignore'):
            icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
        stopped = stopped | (icdist < 0)
        dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
        dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
        x = np.where(stopped, x0, (x0 - dx) * icdist)
        y = np.where(stopped, y0, (y0 - dy) * icdist)
    return np.vstack((x, y))


def undist_point(uv, K, d):
    """Camera.undist_point (common.py:1147-1157): undistort, then re-apply K."""
    xn = undistort_normalized(uv, K, d)
    hom = np.vstack((xn, np.ones(xn.shape[1])))            # util.homogeneous (util.py:54)
    return np.dot(K, hom)[:2]


# ----------------------------------------------------------------------------------------
# tools/util.py
# ----------------------------------------------------------------------------------------
def find_intervals(x, gap=5, idx=False):
    """util.find_intervals (util.py:58-87)."""
    x = np.asarray(x, dtype=np.float64)
    assert x.ndim == 1 and (x[1:] > x[:-1]).all(), 'Input must be an ascending 1D-array'
    x_s, x_e = np.append(-np.inf, x), np.append(x, np.inf)
    start = x_s[1:] - x_s[:-1] >= gap
    end = x_e[:-1] - x_e[1:] <= -gap
    interval = np.array([x[start], x[end]])
    int_idx = np.array([np.where(start)[0], np.where(end)[0]])
    mask = interval[1] - interval[0] >= gap
    interval, int_idx = interval[:, mask], int_idx[:, mask]
    return (interval, int_idx) if idx else interval


def sampling_idx(timestamp, interval):
    """util.sampling(..., belong=True)[1] (util.py:90-116): half-open membership, 0 = none."""
    idx_ts = np.zeros(timestamp.shape, dtype=np.int64)
    for i in range(interval.shape[1]):
        mask = np.logical_xor(timestamp - interval[0, i] >= 0, timestamp - interval[1, i] >= 0)
        idx_ts[mask] = i + 1
    return idx_ts


# ----------------------------------------------------------------------------------------
# FITPACK splev (scipy.interpolate.splev, call sites common.py:294,331)
# ----------------------------------------------------------------------------------------
def _span(t, x):
    n = t.size
    l = np.searchsorted(t, x, side='right') - 1
    return np.clip(l, 3, n - 5)


def _fpbspl(t, x, l):
    """FITPACK fpbspl.f: the 4 non-zero cubic B-splines on span l (de Boor-Cox)."""
    h = np.zeros((4, x.size))
    h[0] = 1.0
    for j in range(1, 4):
        hh = h.copy()
        h[:] = 0.0
        for i in range(j):
            li = l + i + 1
            lj = li - j
            f = hh[i] / (t[li] - t[lj])
            h[i] = h[i] + f * (t[li] - x)
            h[i + 1] = f * (x - t[lj])
    return h


def splev3(x, tck):
    """np.asarray(interpolate.splev(x, tck)) for a 3-D cubic spline, ext=0."""
    t = np.asarray(tck[0], dtype=np.float64)
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    l = _span(t, x)
    h = _fpbspl(t, x, l)
    out = np.zeros((3, x.size))
    for d in range(3):
        c = np.asarray(tck[1][d], dtype=np.float64)
        sp = np.zeros(x.size)
        for q in range(4):
            sp = sp + c[l - 3 + q] * h[q]
        out[d] = sp
    return out


# ----------------------------------------------------------------------------------------
# Problem = the Scene state Scene.BA closes over (common.py:441-697), cameras already in
# ``self.sequence[:numCam]`` order.
# ----------------------------------------------------------------------------------------
@dataclass
class Problem:
    detections: list              # per camera float64[3, M_c]: frame, x_raw, y_raw (common.py:1190)
    H: np.ndarray                 # resolution[1] per camera (common.py:125)
    K: np.ndarray                 # float64[C,3,3]  (fixed unless opt_calib)
    d: np.ndarray                 # float64[C,5]
    knots: list                   # per spline: knot vector t (len n_s + 4)
    interval: np.ndarray          # float64[2,S]
    opt_calib: bool = False
    undist_points: bool = True
    rs: bool = False              # only changes the sparsity pattern (common.py:518-521)
    motion_reg: bool = False
    motion_type: str = 'F'
    motion_weights: float = 1.0
    rs_bounds: bool = False
    opt_sync: bool = True         # settings['opt_sync'] (absent -> the except branch, common.py:512-515): False removes alpha, beta from the pattern

    @property
    def C(self):
        return len(self.detections)

    @property
    def P(self):
        return 15 if self.opt_calib else 6

    @property
    def n_coef(self):
        return [int(np.asarray(t).size - 4) for t in self.knots]

    @property
    def n_params(self):
        return self.C * (3 + self.P) + 3 * sum(self.n_coef)

    def spline_offsets(self):
        """idx_spline_sum[0] of common.py:638-649: start offset of each spline in x."""
        off = self.C * (3 + self.P)
        out = []
        for n in self.n_coef:
            out.append(off)
            off += 3 * n
        return out


def pack_x(prob, alpha, beta, rs, cams, coefs):
    """Parameter vector of Scene.BA (common.py:615-650).

    cams: list of dict(K,R,t,d); coefs: per spline [cx,cy,cz]."""
    model_cam = []
    for cam in cams:
        K = cam['K']
        k = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
        r = rotation_to_rvec(cam['R'])
        if prob.opt_calib:                                   # Camera.P2vector common.py:1113-1124
            model_cam.append(np.concatenate((k, r, cam['t'], cam['d'])))
        else:
            model_cam.append(np.concatenate((r, cam['t'])))
    model_spline = [np.ravel(np.asarray(c)) for c in coefs]
    return np.concatenate([np.asarray(alpha, float), np.asarray(beta, float), np.asarray(rs, float)]
                          + model_cam + model_spline)


def unpack_x(prob, x):
    """np.split sections of error_BA (common.py:454-473) -> alpha, beta, rs, cams, tck."""
    C, P = prob.C, prob.P
    alpha, beta, rs = x[:C], x[C:2 * C], x[2 * C:3 * C]
    cams = []
    for i in range(C):
        v = x[3 * C + i * P: 3 * C + (i + 1) * P]
        if prob.opt_calib:                                   # Camera.vector2P common.py:1127-1144
            K = np.diag((1.0, 1.0, 1.0))
            K[0, 0], K[1, 1] = v[0], v[1]
            K[:2, -1] = v[2:4]
            R = rodrigues(v[4:7])
            t = v[7:10]
            d = v[10:]
        else:
            K, d = prob.K[i], prob.d[i]
            R = rodrigues(v[:3])
            t = v[3:6]
        Pm = np.dot(K, np.hstack((R, t.reshape((-1, 1)))))   # Camera.compose common.py:1082
        cams.append(dict(K=K, R=R, t=t, d=d, P=Pm))
    tck = []
    off = C * (3 + P)
    for s, n in enumerate(prob.n_coef):
        part = x[off: off + 3 * n].reshape(3, -1)
        tck.append([np.asarray(prob.knots[s], float), [part[0], part[1], part[2]], 3])
        off += 3 * n
    return alpha, beta, rs, cams, tck


def detection_to_global(prob, i, alpha, beta, rs, cam):
    """Scene.detection_to_global for one camera (common.py:124-127)."""
    det = prob.detections[i]
    timestamp = alpha[i] * (det[0] + rs[i] * det[2] / prob.H[i]) + beta[i]
    detect = undist_point(det[1:], cam['K'], cam['d']) if prob.undist_points else det[1:]
    return np.vstack((timestamp, detect))


def error_cam_each(prob, i, alpha, beta, rs, cam, tck):
    """Scene.error_cam(cam_id, mode='each') (common.py:304-359), motion_prior=False branch."""
    dg = detection_to_global(prob, i, alpha, beta, rs, cam)
    idx = sampling_idx(dg[0], prob.interval)
    detect = np.empty([3, 0])
    point_3D = np.empty([3, 0])
    for s in range(prob.interval.shape[1]):
        part = dg[:, idx == s + 1]
        if part.size:
            detect = np.hstack((detect, part))
            point_3D = np.hstack((point_3D, splev3(part[0], tck[s])))
    X = np.vstack((point_3D, np.ones(point_3D.shape[1])))
    x = detect[1:]
    x_cal = np.dot(cam['P'], X)                               # Camera.projectPoint common.py:1072-1079
    x_cal = x_cal / x_cal[2]
    error_x = np.zeros_like(prob.detections[i][0])
    error_y = np.zeros_like(prob.detections[i][0])
    error_x[idx.astype(bool)] = abs(x_cal[0] - x[0])
    error_y[idx.astype(bool)] = abs(x_cal[1] - x[1])
    return np.concatenate((error_x, error_y))


def spline_to_traj(prob, tck, sampling_rate=1):
    """Scene.spline_to_traj() with t=None (common.py:273-301)."""
    interval = prob.interval
    traj = np.empty([4, 0])
    timestamp = np.arange(interval[0, 0], interval[1, -1], sampling_rate)
    for s in range(interval.shape[1]):
        t_part = timestamp[np.logical_and(timestamp >= interval[0, s], timestamp <= interval[1, s])]
        traj_part = splev3(t_part, tck[s]) if t_part.size else np.empty([3, 0])
        traj = np.hstack((traj, np.vstack((t_part, traj_part))))
    return traj


def motion_prior(traj, weights, eps=1e-20, prior='F'):
    """Scene.motion_prior (common.py:959-1001)."""
    ts = traj[0]
    if prior == 'KE':
        traj_for = traj[1:, :-1]
        traj_aft = traj[1:, 1:]
        vel = (traj_aft - traj_for) / ((ts[1:] - ts[:-1]) + eps)
        mot_resid = np.array([weights[:traj_for.shape[1]] * 0.5 * (vel ** 2 * (ts[1:] - ts[:-1]))])
    if prior == 'F':
        traj_for = traj[1:, :-2]
        traj_mid = traj[1:, 1:-1]
        traj_aft = traj[1:, 2:]
        dt1 = ts[1:-1] - ts[:-2]
        dt2 = ts[2:] - ts[1:-1]
        dt3 = dt1 + dt2
        v1 = (traj_mid - traj_for) / (dt1 + eps)
        v2 = (traj_aft - traj_mid) / (dt2 + eps)
        accel = (v2 - v1) / (dt3 + eps)
        mot_resid = np.array([weights[:traj_for.shape[1]] * (accel * (dt3))])
    return np.sum(abs(mot_resid[0]), axis=0)


def error_motion_reg(prob, tck):
    """Scene.error_motion(..., motion_reg=True) (common.py:362-424), motion_reg branch."""
    traj = spline_to_traj(prob, tck)
    idx = sampling_idx(traj[0], prob.interval)
    traj_ts = np.array([])
    mot_err_res = np.array([])
    motion_error = np.zeros((traj.shape[1]))
    for s in range(prob.interval.shape[1]):
        traj_part = traj[:, idx == s + 1]
        if traj_part.size:
            weights = np.ones(traj_part.shape[1]) * prob.motion_weights
            mot_err = motion_prior(traj_part, weights, prior=prob.motion_type)
            mot_err_res = np.concatenate((mot_err_res, mot_err))
            if prob.motion_type == 'F':
                traj_ts = np.concatenate((traj_ts, traj_part[0, 1:-1]))
            elif prob.motion_type == 'KE':
                traj_ts = np.concatenate((traj_ts, traj_part[0, 1:]))
    _, traj_idx, _ = np.intersect1d(traj[0], traj_ts, assume_unique=True, return_indices=True)
    motion_error[traj_idx] = mot_err_res
    return motion_error


def residual(prob, x):
    """error_BA (common.py:448-487), motion_prior=False."""
    alpha, beta, rs, cams, tck = unpack_x(prob, np.asarray(x, dtype=np.float64))
    error = np.array([])
    for i in range(prob.C):
        error = np.concatenate((error, error_cam_each(prob, i, alpha, beta, rs, cams[i], tck)))
    if prob.motion_reg:
        error = np.concatenate((error, error_motion_reg(prob, tck)))
    return error


def jac_pattern(prob, x0, near=3):
    """jac_BA (common.py:490-610): the sparsity pattern handed to least_squares, as CSR.

    Only the motion_prior=False branch; returns a scipy.sparse.csr_matrix of int
    (the reference returns the same matrix dense, common.py:610).

    Note on ties: ``knot = t[2:-2]`` repeats the interval start and end twice, so for a
    timestamp in the first/last knot span two candidates can be exactly equidistant and
    ``np.argsort`` (default, unstable kind) decides which one enters the top three.  The
    call is kept verbatim; its tie order depends on numpy's sort kernel for the array
    length / CPU, so only those rows are machine-dependent."""
    from scipy.sparse import csr_matrix
    alpha, beta, rs, cams, tck = unpack_x(prob, np.asarray(x0, dtype=np.float64))
    C, P, n = prob.C, prob.P, prob.n_params
    offs = prob.spline_offsets()
    rows, cols = [], []
    row0 = 0
    for i in range(C):
        dg = detection_to_global(prob, i, alpha, beta, rs, cams[i])
        visible = sampling_idx(dg[0], prob.interval)             # compute_visibility common.py:427-438
        M = dg.shape[1]
        r_cam, c_cam = [], []
        vis_rows = np.nonzero(visible)[0]
        cam_cols = ([i, i + C] if prob.opt_sync else []) + ([i + 2 * C] if prob.rs else []) + list(range(3 * C + i * P, 3 * C + (i + 1) * P))
        for cc in cam_cols:
            r_cam.append(vis_rows)
            c_cam.append(np.full(vis_rows.size, cc))
        for j in vis_rows:
            s = visible[j] - 1
            knot = np.asarray(prob.knots[s], float)[2:-2]
            knot_idx = np.argsort(abs(knot - dg[0, j]))[:near]          # verbatim common.py:561, see note
            knot_idx = np.concatenate((knot_idx, knot_idx + len(knot), knot_idx + 2 * len(knot)))
            r_cam.append(np.full(knot_idx.size, j))
            c_cam.append(offs[s] + knot_idx)
        r_cam = np.concatenate(r_cam) if r_cam else np.array([], int)
        c_cam = np.concatenate(c_cam) if c_cam else np.array([], int)
        rows += [row0 + r_cam, row0 + M + r_cam]                  # vstack([jac_cam, jac_cam]) common.py:568
        cols += [c_cam, c_cam]
        row0 += 2 * M
    if prob.motion_reg:
        traj = spline_to_traj(prob, tck)
        S = prob.interval.shape[1]
        for j in range(traj.shape[1]):
            sid = int(sampling_idx(traj[0, j:j + 1], prob.interval)[0]) - 1
            if sid < 0:
                sid = S - 1                                       # python negative index, common.py:579-584
            knot = np.asarray(prob.knots[sid], float)[2:-2]
            knot_idx = np.argsort(abs(knot - traj[0, j]))[:near]         # verbatim common.py:582
            knot_idx = np.concatenate((knot_idx, knot_idx + len(knot), knot_idx + 2 * len(knot)))
            rows.append(np.full(knot_idx.size, row0 + j))
            cols.append(offs[sid] + knot_idx)
        row0 += traj.shape[1]
    rows = np.concatenate(rows).astype(np.int64)
    cols = np.concatenate(cols).astype(np.int64)
    A = csr_matrix((np.ones(rows.size, dtype=np.int64), (rows, cols)), shape=(row0, n))
    A.sum_duplicates()
    A.data[:] = 1
    return A


def bounds(prob):
    """rs bounds of Scene.BA (common.py:654-662)."""
    n, C = prob.n_params, prob.C
    if prob.rs_bounds:
        lb = np.ones(n) * -np.inf
        ub = np.ones(n) * np.inf
        lb[2 * C:3 * C] = 0
        ub[2 * C:3 * C] = 1
        return (lb, ub)
    return (-np.inf, np.inf)


def solve(prob, x0, max_iter=10, pattern=None):
    """The least_squares call of Scene.BA (common.py:665-670), verbatim arguments."""
    from scipy.optimize import least_squares
    A = jac_pattern(prob, x0) if pattern is None else pattern
    fn = lambda x: residual(prob, x)
    return least_squares(fn, np.asarray(x0, float), jac_sparsity=A, tr_solver='lsmr', xtol=1e-12,
                         max_nfev=max_iter, verbose=0, bounds=bounds(prob))


def outlier_keep_mask(prob, x, thres):
    """Scene.remove_outliers (common.py:700-717): per camera ``error < thres``."""
    alpha, beta, rs, cams, tck = unpack_x(prob, np.asarray(x, dtype=np.float64))
    masks = []
    for i in range(prob.C):
        error_all = error_cam_each(prob, i, alpha, beta, rs, cams[i], tck)
        error_xy = np.split(error_all, 2)
        error = np.sqrt(error_xy[0] ** 2 + error_xy[1] ** 2)
        masks.append(error < thres)
    return masks


def reprojection_rmse(prob, x):
    """sqrt(mean(error_cam(i,'dist')^2)) over visible detections."""
    f = residual(prob, x)
    sq, cnt, off = 0.0, 0, 0
    for i in range(prob.C):
        M = prob.detections[i].shape[1]
        ex, ey = f[off:off + M], f[off + M:off + 2 * M]
        vis = (ex != 0) | (ey != 0)
        sq += float(np.sum(ex[vis] ** 2 + ey[vis] ** 2))
        cnt += int(vis.sum())
        off += 2 * M
    return float(np.sqrt(sq / max(cnt, 1)))


def numeric_jacobian(prob, x, cols=None, rel=1e-6):
    """Central finite differences of :func:`residual` (dense, test helper; not the reference's FD)."""
    x = np.asarray(x, dtype=np.float64)
    cols = range(x.size) if cols is None else cols
    f0 = residual(prob, x)
    J = np.zeros((f0.size, len(list(cols))))
    for k, j in enumerate(cols):
        h = rel * max(1.0, abs(x[j]))
        xp, xm = x.copy(), x.copy()
        xp[j] += h
        xm[j] -= h
        J[:, k] = (residual(prob, xp) - residual(prob, xm)) / (xp[j] - xm[j])
    return J


def problem_from_scene(scene, num_cam=None, rs=None, motion_reg=None, motion_weights=None, rs_bounds=None):
    """Build a Problem + x0 from a synthetic scene object with the reference's attribute names
    (cameras[i] dicts K,d,R,t,resolution; detections; alpha; beta; rs; tck; interval; settings)."""
    C = scene.num_cam if num_cam is None else num_cam
    st = scene.settings
    prob = Problem(
        detections=[np.asarray(scene.detections[i], float) for i in range(C)],
        H=np.array([scene.cameras[i]['resolution'][1] for i in range(C)], float),
        K=np.array([scene.cameras[i]['K'] for i in range(C)], float),
        d=np.array([scene.cameras[i]['d'] for i in range(C)], float),
        knots=[np.asarray(t[0], float) for t in scene.tck],
        interval=np.asarray(scene.interval, float),
        opt_calib=bool(st.get('opt_calib', False)),
        undist_points=bool(st.get('undist_points', True)),
        rs=bool(st.get('rolling_shutter', False) if rs is None else rs),
        motion_reg=bool(st.get('motion_reg', False) if motion_reg is None else motion_reg),
        motion_type=st.get('motion_type', 'F'),
        motion_weights=float(st.get('motion_weights', 1.0) if motion_weights is None else motion_weights),
        rs_bounds=bool(st.get('rs_bounds', False) if rs_bounds is None else rs_bounds),
        opt_sync=bool(st.get('opt_sync', True)),
    )
    x0 = pack_x(prob, scene.alpha[:C], scene.beta[:C], scene.rs[:C],
                [scene.cameras[i] for i in range(C)], [t[1] for t in scene.tck])
    return prob, x0
