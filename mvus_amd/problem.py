"""Flat description of one ``Scene.BA`` problem, in the layout the C ABI (include/mvus_ba.h) takes.

This is the host-side codec between the reference's ``Scene`` state and the arrays the HIP kernels
read.  The parameter-vector layout is exactly the reference's (``reconstruction/common.py:615-650``):

    x = [alpha(C) ; beta(C) ; rs(C) ; cam_0(P) ... cam_{C-1}(P) ; spline_0: cx,cy,cz ; spline_1 ...]

with cameras in ``sequence[:numCam]`` order, ``P = 6`` (rvec, t) or ``15``
(fx, fy, cx, cy, rvec, t, k1, k2, p1, p2, k3) when ``opt_calib`` (``Camera.P2vector``, common.py:1113-1124).
"""
from dataclasses import dataclass
import numpy as np

from .synth import rodrigues, rotation_to_rvec  # cv2.Rodrigues stand-ins (host-side codec only)

MOTION_F = 0
MOTION_KE = 1


@dataclass
class BAProblem:
    num_cam: int
    opt_calib: bool
    undist_points: bool
    rs_free: bool               # ``rs`` argument of Scene.BA: is the rolling-shutter column optimised
    rs_bounds: bool
    motion_reg: bool
    motion_type: int            # MOTION_F / MOTION_KE
    motion_weight: float
    det_offsets: np.ndarray     # int64[C+1]
    frame: np.ndarray           # float64[M]  camera-segmented
    u_raw: np.ndarray
    v_raw: np.ndarray
    img_height: np.ndarray      # float64[C]  resolution[1]
    K: np.ndarray               # float64[C,4] fx fy cx cy (fixed unless opt_calib)
    dist: np.ndarray            # float64[C,5]
    interval: np.ndarray        # float64[2,S]
    knot_offsets: np.ndarray    # int64[S+1]
    knots: np.ndarray           # float64[sum(n_s+4)]
    opt_sync: bool = True       # settings['opt_sync'] (absent = True): False freezes alpha and beta (common.py:512-515)

    @property
    def C(self):
        return self.num_cam

    @property
    def P(self):
        return 15 if self.opt_calib else 6

    @property
    def M(self):
        return int(self.det_offsets[-1])

    @property
    def S(self):
        return int(self.interval.shape[1])

    @property
    def n_coef(self):
        return (np.diff(self.knot_offsets) - 4).astype(np.int64)

    @property
    def n_params(self):
        return int(self.C * (3 + self.P) + 3 * self.n_coef.sum())

    @property
    def spline_x_offsets(self):
        """idx_spline_sum[0] of common.py:638-649."""
        base = self.C * (3 + self.P)
        return base + 3 * np.concatenate(([0], np.cumsum(self.n_coef)[:-1])).astype(np.int64)

    @property
    def ctrl_offsets(self):
        return np.concatenate(([0], np.cumsum(self.n_coef))).astype(np.int64)

    def motion_sample_times(self):
        """Timestamps of Scene.spline_to_traj() (common.py:289-297): arange over the whole span,
        kept per interval where start <= t <= end.  Returns (ts, interval_id)."""
        iv = self.interval
        grid = np.arange(iv[0, 0], iv[1, -1], 1)
        ts, sid = [], []
        for s in range(iv.shape[1]):
            part = grid[np.logical_and(grid >= iv[0, s], grid <= iv[1, s])]
            ts.append(part)
            sid.append(np.full(part.size, s, dtype=np.int32))
        return np.concatenate(ts), np.concatenate(sid)

    @property
    def num_motion_rows(self):
        return int(self.motion_sample_times()[0].size) if self.motion_reg else 0

    @property
    def n_residuals(self):
        return 2 * self.M + self.num_motion_rows

    def bounds(self):
        """rs bounds of Scene.BA (common.py:654-662)."""
        lb = np.full(self.n_params, -np.inf)
        ub = np.full(self.n_params, np.inf)
        if self.rs_bounds:
            lb[2 * self.C:3 * self.C] = 0.0
            ub[2 * self.C:3 * self.C] = 1.0
        return lb, ub

    def shard(self, rank, world):
        """Observation shard ``rank`` of ``world``: every camera's detections are split into ``world``
        contiguous (time-ordered) pieces; piece ``rank`` of every camera goes to this rank.  Camera
        parameters and the spline stay replicated; motion rows belong to rank 0 .. see sharding.py."""
        from .sharding import shard_offsets
        keep = []
        new_off = [0]
        for c in range(self.C):
            a, b = int(self.det_offsets[c]), int(self.det_offsets[c + 1])
            lo, hi = shard_offsets(b - a, rank, world)
            keep.append(np.arange(a + lo, a + hi))
            new_off.append(new_off[-1] + (hi - lo))
        keep = np.concatenate(keep) if keep else np.zeros(0, dtype=np.int64)
        return BAProblem(
            num_cam=self.num_cam, opt_calib=self.opt_calib, undist_points=self.undist_points,
            rs_free=self.rs_free, rs_bounds=self.rs_bounds, motion_reg=self.motion_reg,
            motion_type=self.motion_type, motion_weight=self.motion_weight,
            det_offsets=np.asarray(new_off, dtype=np.int64), frame=self.frame[keep].copy(),
            u_raw=self.u_raw[keep].copy(), v_raw=self.v_raw[keep].copy(), img_height=self.img_height,
            K=self.K, dist=self.dist, interval=self.interval, knot_offsets=self.knot_offsets,
            knots=self.knots, opt_sync=self.opt_sync), keep


    # ---- time sharding --------------------------------------------------------------------------
    def detection_spans(self, x):
        """Global index of the first of the four control points each detection touches at parameters ``x``
        (-1 = outside every spline interval), the quantity time shards are cut by."""
        C = self.C
        alpha, beta, rs = x[:C], x[C:2 * C], x[2 * C:3 * C]
        from .bspline import find_span
        ctrl_off = np.concatenate(([0], np.cumsum(self.n_coef))).astype(np.int64)
        out = np.full(self.M, -1, dtype=np.int64)
        for c in range(C):
            a, b = int(self.det_offsets[c]), int(self.det_offsets[c + 1])
            tau = alpha[c] * (self.frame[a:b] + rs[c] * self.v_raw[a:b] / self.img_height[c]) + beta[c]     # common.py:125, rs fixed or free
            for s_ in range(self.S):
                t = self.knots[int(self.knot_offsets[s_]):int(self.knot_offsets[s_ + 1])]
                inside = (tau >= self.interval[0, s_]) & (tau < self.interval[1, s_])
                if inside.any():
                    out[a:b][inside] = ctrl_off[s_] + find_span(t, tau[inside]) - 3
        return out

    def time_cuts(self, x, world, halo=8):
        """Control-point cuts [0, c_1, .., N] giving every rank about the same number of detections."""
        g = self.detection_spans(x)
        g = np.sort(g[g >= 0])
        N = int(np.sum(self.n_coef))
        cuts = [0]
        for r in range(1, world):
            cuts.append(int(g[min(g.size - 1, (g.size * r) // world)]) if g.size else (N * r) // world)
        cuts.append(N)
        gap = 2 * halo + 8                               # the library needs room for the halo and one separator per rank
        for r in range(1, world):
            cuts[r] = min(max(cuts[r], cuts[r - 1] + gap), N - gap * (world - r))
        if any(cuts[r + 1] - cuts[r] < gap for r in range(world)):
            raise ValueError('too few control points (%d) for %d time shards with halo %d' % (N, world, halo))
        return np.asarray(cuts, dtype=np.int32)

    def shard_time(self, rank, world, x, halo=8, cuts=None):
        """Time shard ``rank`` of ``world``: the detections whose first control point (at parameters ``x``) lies in
        [cuts[rank], cuts[rank+1]); detections outside every interval follow their predecessor in the camera.
        Returns (sub-problem, kept detection indices, cuts).  Parameters, splines and motion samples stay replicated."""
        if cuts is None:
            cuts = self.time_cuts(x, world, halo)
        g = self.detection_spans(x)
        keep, new_off = [], [0]
        for c in range(self.C):
            a, b = int(self.det_offsets[c]), int(self.det_offsets[c + 1])
            gc = g[a:b]
            src = np.maximum.accumulate(np.where(gc >= 0, np.arange(gc.size), -1))     # forward fill the invisible ones
            gc = np.where(src >= 0, gc[np.maximum(src, 0)], 0)
            owner = np.searchsorted(cuts[1:-1], gc, side='right')
            idx = a + np.nonzero(owner == rank)[0]
            keep.append(idx)
            new_off.append(new_off[-1] + idx.size)
        keep = np.concatenate(keep) if keep else np.zeros(0, dtype=np.int64)
        sub = BAProblem(
            num_cam=self.num_cam, opt_calib=self.opt_calib, undist_points=self.undist_points,
            rs_free=self.rs_free, rs_bounds=self.rs_bounds, motion_reg=self.motion_reg,
            motion_type=self.motion_type, motion_weight=self.motion_weight,
            det_offsets=np.asarray(new_off, dtype=np.int64), frame=self.frame[keep].copy(),
            u_raw=self.u_raw[keep].copy(), v_raw=self.v_raw[keep].copy(), img_height=self.img_height,
            K=self.K, dist=self.dist, interval=self.interval, knot_offsets=self.knot_offsets,
            knots=self.knots, opt_sync=self.opt_sync)
        return sub, keep, cuts


def problem_from_arrays(detections, cameras, tck, interval, *, opt_calib=False, undist_points=True,
                        rs=False, rs_bounds=False, motion_reg=False, motion_type='F', motion_weights=1.0, opt_sync=True):
    """Build a BAProblem from the reference's containers: ``detections[i]`` float64[3,M_i] rows
    (frame, x, y); ``cameras[i]`` with K, d, resolution (dicts or objects); ``tck`` as in
    ``Scene.spline['tck']``; ``interval`` = ``Scene.spline['int']``.  Cameras are already in BA order."""
    get = (lambda c, k: c[k]) if isinstance(cameras[0], dict) else (lambda c, k: getattr(c, k))
    C = len(cameras)
    det_off = np.concatenate(([0], np.cumsum([d.shape[1] for d in detections]))).astype(np.int64)
    det = np.hstack([np.asarray(d, dtype=np.float64) for d in detections]) if C else np.zeros((3, 0))
    K = np.array([[get(c, 'K')[0, 0], get(c, 'K')[1, 1], get(c, 'K')[0, 2], get(c, 'K')[1, 2]] for c in cameras],
                 dtype=np.float64)
    dist = np.array([np.asarray(get(c, 'd'), dtype=np.float64).reshape(-1)[:5] for c in cameras], dtype=np.float64)
    H = np.array([get(c, 'resolution')[1] for c in cameras], dtype=np.float64)
    knots = [np.asarray(t[0], dtype=np.float64) for t in tck]
    koff = np.concatenate(([0], np.cumsum([k.size for k in knots]))).astype(np.int64)
    if motion_type not in ('F', 'KE'):
        raise AssertionError('Motion type must be either F or KE')     # common.py:416
    return BAProblem(
        num_cam=C, opt_calib=bool(opt_calib), undist_points=bool(undist_points), rs_free=bool(rs),
        rs_bounds=bool(rs_bounds), motion_reg=bool(motion_reg),
        motion_type=MOTION_F if motion_type == 'F' else MOTION_KE, motion_weight=float(motion_weights),
        det_offsets=det_off, frame=np.ascontiguousarray(det[0]), u_raw=np.ascontiguousarray(det[1]),
        v_raw=np.ascontiguousarray(det[2]), img_height=H, K=K, dist=dist,
        interval=np.ascontiguousarray(np.asarray(interval, dtype=np.float64)),
        knot_offsets=koff, knots=np.concatenate(knots) if knots else np.zeros(0), opt_sync=bool(opt_sync))


def pack_x(prob, alpha, beta, rs, cameras, tck):
    """Scene.BA parameter packing (common.py:615-650)."""
    get = (lambda c, k: c[k]) if isinstance(cameras[0], dict) else (lambda c, k: getattr(c, k))
    parts = [np.asarray(alpha, dtype=np.float64), np.asarray(beta, dtype=np.float64), np.asarray(rs, dtype=np.float64)]
    for c in cameras:
        K = get(c, 'K')
        r = rotation_to_rvec(get(c, 'R'))
        t = np.asarray(get(c, 't'), dtype=np.float64).reshape(3)
        if prob.opt_calib:
            parts.append(np.concatenate(([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], r, t,
                                         np.asarray(get(c, 'd'), dtype=np.float64).reshape(-1)[:5])))
        else:
            parts.append(np.concatenate((r, t)))
    for t in tck:
        parts.append(np.ravel(np.asarray(t[1], dtype=np.float64)))
    return np.concatenate(parts)


def unpack_x(prob, x):
    """Inverse of :func:`pack_x` (common.py:672-692): alpha, beta, rs, per-camera dict(K,R,t,d), coefficient list."""
    x = np.asarray(x, dtype=np.float64)
    C, P = prob.C, prob.P
    alpha, beta, rs = x[:C].copy(), x[C:2 * C].copy(), x[2 * C:3 * C].copy()
    cams = []
    for i in range(C):
        v = x[3 * C + i * P:3 * C + (i + 1) * P]
        if prob.opt_calib:
            K = np.diag((1.0, 1.0, 1.0))
            K[0, 0], K[1, 1] = v[0], v[1]
            K[:2, -1] = v[2:4]
            cams.append(dict(K=K, R=rodrigues(v[4:7]), t=v[7:10].copy(), d=v[10:15].copy()))
        else:
            K = np.diag((1.0, 1.0, 1.0))
            K[0, 0], K[1, 1], K[0, 2], K[1, 2] = prob.K[i]
            cams.append(dict(K=K, R=rodrigues(v[:3]), t=v[3:6].copy(), d=prob.dist[i].copy()))
    coefs = []
    off = C * (3 + P)
    for n in prob.n_coef:
        part = x[off:off + 3 * n].reshape(3, -1)
        coefs.append([part[0].copy(), part[1].copy(), part[2].copy()])
        off += 3 * n
    return alpha, beta, rs, cams, coefs


def problem_from_scene(scene, num_cam=None, **overrides):
    """BAProblem + x0 from a scene-like object (synth.SynthScene or anything with the same fields)."""
    C = scene.num_cam if num_cam is None else num_cam
    st = dict(scene.settings)
    st.update(overrides)
    prob = problem_from_arrays(
        [scene.detections[i] for i in range(C)], [scene.cameras[i] for i in range(C)], scene.tck, scene.interval,
        opt_calib=st.get('opt_calib', False), undist_points=st.get('undist_points', True),
        rs=st.get('rolling_shutter', False), rs_bounds=st.get('rs_bounds', False),
        motion_reg=st.get('motion_reg', False), motion_type=st.get('motion_type', 'F'),
        motion_weights=st.get('motion_weights', 1.0), opt_sync=st.get('opt_sync', True))
    x0 = pack_x(prob, scene.alpha[:C], scene.beta[:C], scene.rs[:C], [scene.cameras[i] for i in range(C)], scene.tck)
    return prob, x0
