"""Seeded synthetic multi-camera scenes for the BA hot path.

The reference ships no data; the generator below follows the
recipe of SURVEY.md section 8(d): a smooth closed-form 3-D curve, cameras on a ring
looking at its centroid, unsynchronised frame rates / time shifts, optional rolling
shutter and lens distortion, pixel noise and gross outliers, optional visibility
gaps so that the trajectory has several spline intervals.

The output mirrors the state the reference ``Scene`` holds when ``Scene.BA`` is
entered (reference ``reconstruction/common.py:38-61,441``): per-camera detections
``(frame, x, y)``, ``alpha/beta/rs``, cameras ``K,d,R,t``, and the spline
``{'tck': [[t,[cx,cy,cz],3],...], 'int': 2xS}``.
"""
from dataclasses import dataclass, field
import numpy as np

from . import bspline


def rodrigues(rvec):
    """Rotation vector -> 3x3 matrix (what cv2.Rodrigues computes, reference common.py:1136)."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    th = np.sqrt(r @ r)
    if th < 2.220446049250313e-16:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    c, s = np.cos(th), np.sin(th)
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * Kx


def rotation_to_rvec(R):
    """3x3 rotation -> rotation vector (inverse of :func:`rodrigues`, reference common.py:1119)."""
    R = np.asarray(R, dtype=np.float64)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = 0.5 * np.sqrt(w @ w)
    c = np.clip(0.5 * (np.trace(R) - 1.0), -1.0, 1.0)
    th = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        # theta ~ pi: (R + I)/2 = k k^T, take the best-conditioned column
        B = 0.5 * (R + np.eye(3))
        i = int(np.argmax(np.diag(B)))
        ax = B[:, i] / np.sqrt(B[i, i])
        return ax * (th / np.sqrt(ax @ ax))
    return w * (0.5 * th / s)


def curve(t):
    """The closed-form target trajectory of SURVEY.md 8(d)."""
    t = np.asarray(t, dtype=np.float64)
    return np.vstack((10.0 * np.sin(t / 80.0), 10.0 * np.cos(t / 95.0), 30.0 + 3.0 * np.sin(t / 50.0)))


def look_at(center, target):
    z = target - center
    z = z / np.linalg.norm(z)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(z, up)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.vstack((x, y, z))
    return R, -R @ center


def distort(xn, yn, d):
    """Forward 5-coefficient lens model (k1,k2,p1,p2,k3) on normalised coordinates."""
    k1, k2, p1, p2, k3 = d
    r2 = xn * xn + yn * yn
    rad = 1.0 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = xn * rad + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    yd = yn * rad + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
    return xd, yd


@dataclass
class SynthScene:
    cameras: list            # dicts: K(3x3) d(5) R(3x3) t(3) fps resolution[W,H]
    detections: list         # per camera float64[3, M_c]: rows frame, x, y
    alpha: np.ndarray
    beta: np.ndarray
    rs: np.ndarray
    tck: list                # [[t, [cx,cy,cz], 3], ...]
    interval: np.ndarray     # float64[2, S]
    settings: dict
    truth: dict = field(default_factory=dict)

    @property
    def num_cam(self):
        return len(self.cameras)

    @property
    def num_obs(self):
        return int(sum(d.shape[1] for d in self.detections))


def make_scene(num_cam, total_obs, *, seed=0, rolling_shutter=False, distortion=False,
               opt_calib=False, motion_reg=False, motion_type='F', motion_weights=1.0,
               rs_bounds=False, num_knots=None, knot_spacing=15.0, num_intervals=1,
               noise_px=0.5, outlier_frac=0.02, dropout=0.0, perturb=1.0,
               fps_choices=(25.0, 30.0, 50.0, 59.94), ring_radius=60.0):
    """Build a seeded scene with ``num_cam`` cameras and about ``total_obs`` detections.

    ``ring_radius`` (m) is the distance of the camera ring from the trajectory's centroid: at the default 60 m the target
    stays within ~170 px of the image centre (lens distortion is then barely observable -- fine while calibration is fixed);
    ~22 m makes it sweep most of the image, which is what a scene with ``opt_calib`` needs to be well posed."""
    rng = np.random.default_rng(seed)
    C = num_cam
    fps = np.array([fps_choices[i % len(fps_choices)] for i in range(C)], dtype=np.float64)
    fps[0] = fps_choices[1 % len(fps_choices)]
    alpha_true = fps[0] / fps
    beta_true = rng.uniform(-50.0, 50.0, C)
    beta_true[0] = 0.0
    rs_true = rng.uniform(0.1, 0.9, C) if rolling_shutter else np.zeros(C)

    # global duration (reference-camera frames) giving ~total_obs detections
    G = float(np.ceil(total_obs / np.sum(1.0 / alpha_true) / max(1.0 - dropout, 1e-3))) + 110.0
    if num_knots is not None:
        knot_spacing = max(G / num_knots, 1.5)

    # spline intervals with gaps of 12 reference frames between them
    gap = 12.0
    edges = np.linspace(0.0, G, num_intervals + 1)
    interval = np.zeros((2, num_intervals))
    for s in range(num_intervals):
        interval[0, s] = edges[s] + (gap / 2 if s > 0 else 0.0) + 0.37
        interval[1, s] = edges[s + 1] - (gap / 2 if s < num_intervals - 1 else 0.0) - 0.41

    tck_true = []
    for s in range(num_intervals):
        a, b = interval[:, s]
        t = bspline.make_knots(a, b, knot_spacing, rng)
        xs = np.linspace(a, b, max(int((b - a) * 2), 8 * (t.size - 4)))
        c = bspline.lsq_fit(t, xs, curve(xs))
        tck_true.append([t, [c[0].copy(), c[1].copy(), c[2].copy()], 3])

    centroid = np.array([0.0, 0.0, 30.0])
    cameras, detections = [], []
    W, H = 1920, 1080
    for c in range(C):
        ang = 2 * np.pi * c / C + rng.uniform(-0.1, 0.1)
        elev = rng.uniform(-8.0, 8.0)
        center = centroid + np.array([ring_radius * np.cos(ang), ring_radius * np.sin(ang), elev])
        R, tvec = look_at(center, centroid)
        f = 1000.0 + rng.uniform(-50, 50)
        K = np.array([[f, 0, W / 2 + rng.uniform(-10, 10)], [0, f * rng.uniform(0.995, 1.005), H / 2 + rng.uniform(-10, 10)], [0, 0, 1.0]])
        d = (rng.normal(0, 1, 5) * np.array([0.1, 0.01, 1e-4, 1e-4, 0.0])) if distortion else np.zeros(5)
        # make R an exact image of a rotation vector so P2vector/vector2P round-trips
        R = rodrigues(rotation_to_rvec(R))
        cameras.append(dict(K=K, d=d, R=R, t=tvec, fps=float(fps[c]), resolution=[W, H]))

        f0 = int(np.ceil(max((0.0 - beta_true[c]) / alpha_true[c], 0.0)))
        f1 = int(np.floor((G - beta_true[c]) / alpha_true[c]))
        frames = np.arange(f0, f1, dtype=np.float64)
        if dropout > 0:
            frames = frames[rng.uniform(size=frames.size) >= dropout]
        # rolling shutter: the row offset depends on the projected row -> fixed point
        v = np.full(frames.size, H / 2.0)
        for _ in range(4):
            tau = alpha_true[c] * (frames + rs_true[c] * v / H) + beta_true[c]
            X = np.zeros((3, frames.size))
            inside = np.zeros(frames.size, dtype=bool)
            for s in range(num_intervals):
                m = (tau >= interval[0, s]) & (tau < interval[1, s])
                if m.any():
                    X[:, m] = bspline.evaluate(tck_true[s][0], np.array(tck_true[s][1]), tau[m])
                inside |= m
            X[:, ~inside] = curve(tau[~inside])
            Xc = R @ X + tvec[:, None]
            xn, yn = Xc[0] / Xc[2], Xc[1] / Xc[2]
            xd, yd = distort(xn, yn, d)
            u = K[0, 0] * xd + K[0, 2]
            v = K[1, 1] * yd + K[1, 2]
        u = u + rng.normal(0, noise_px, u.size)
        v = v + rng.normal(0, noise_px, v.size)
        out = rng.uniform(size=u.size) < outlier_frac
        r_out = rng.uniform(20, 200, u.size)
        a_out = rng.uniform(0, 2 * np.pi, u.size)
        u = np.where(out, u + r_out * np.cos(a_out), u)
        v = np.where(out, v + r_out * np.sin(a_out), v)
        keep = (u >= 0) & (u < W) & (v >= 0) & (v < H)
        detections.append(np.vstack((frames[keep], u[keep], v[keep])))

    truth = dict(alpha=alpha_true.copy(), beta=beta_true.copy(), rs=rs_true.copy(),
                 cameras=[{k: (np.array(vv, dtype=np.float64).copy() if k in ('K', 'd', 'R', 't') else vv)
                           for k, vv in cam.items()} for cam in cameras],
                 tck=[[t.copy(), [ci.copy() for ci in cs], 3] for t, cs, _ in tck_true])

    # initial state = truth + perturbation (what BA has to undo)
    p = float(perturb)
    alpha = alpha_true.copy()
    beta = beta_true + rng.normal(0, 0.3 * p, C)
    rs = np.clip(rs_true + rng.normal(0, 0.05 * p, C), 0.02, 0.98) if rolling_shutter else np.zeros(C)
    for cam in cameras:
        rv = rotation_to_rvec(cam['R']) + rng.normal(0, 1e-3 * p, 3)
        cam['R'] = rodrigues(rv)
        cam['t'] = cam['t'] + rng.normal(0, 0.03 * p, 3)
        if opt_calib:
            cam['K'] = cam['K'].copy()
            cam['K'][0, 0] += rng.normal(0, 2.0 * p)
            cam['K'][1, 1] += rng.normal(0, 2.0 * p)
            cam['K'][0, 2] += rng.normal(0, 1.0 * p)
            cam['K'][1, 2] += rng.normal(0, 1.0 * p)
    tck = []
    for t, cs, k in tck_true:
        tck.append([t.copy(), [ci + rng.normal(0, 0.01 * p, ci.size) for ci in cs], 3])

    settings = dict(undist_points=True, opt_calib=bool(opt_calib), rolling_shutter=bool(rolling_shutter),
                    rs_bounds=bool(rs_bounds), motion_reg=bool(motion_reg), motion_type=motion_type,
                    motion_weights=float(motion_weights), smooth_factor=[10, 20], thres_outlier=10,
                    ref_cam=0, camera_sequence=list(range(C)))
    return SynthScene(cameras=cameras, detections=detections, alpha=alpha, beta=beta, rs=rs,
                      tck=tck, interval=interval, settings=settings, truth=truth)


# The BASELINE.json configurations (index -> generator arguments).
BASELINE_CONFIGS = {
    0: dict(num_cam=2, total_obs=2000, seed=1, rolling_shutter=False, motion_reg=False, knot_spacing=15.0),
    1: dict(num_cam=7, total_obs=100_000, seed=2, rolling_shutter=True, motion_reg=True, motion_type='F',
            motion_weights=1e4, knot_spacing=15.0, num_intervals=2),
    2: dict(num_cam=32, total_obs=500_000, seed=3, rolling_shutter=True, num_knots=5000),
    3: dict(num_cam=64, total_obs=2_000_000, seed=4, rolling_shutter=True, num_knots=8000),
    4: dict(num_cam=7, total_obs=100_000, seed=5, rolling_shutter=True, distortion=True, opt_calib=True,
            rs_bounds=True, motion_reg=True, motion_type='KE', motion_weights=1e2, knot_spacing=15.0),
}


def baseline_scene(index, **overrides):
    kw = dict(BASELINE_CONFIGS[index])
    kw.update(overrides)
    return make_scene(**kw)
