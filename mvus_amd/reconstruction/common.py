"""``Scene`` / ``Camera`` / ``create_scene`` with the reference's names, attributes and config surface
(reference ``multiviewunsynch/reconstruction/common.py``), with the bundle-adjustment hot path --
``Scene.BA`` (common.py:441-697), ``Scene.remove_outliers`` (:700-717), ``Scene.error_cam`` (:304-359) --
running on the GPU through libmvusba.so.  There is no CPU fallback for those three.

Kept from the reference: the ``Scene`` attribute set that ends up in the output pickle (README "Output"),
method names and signatures, the ``config.json`` schema (``create_scene``), the parameter-vector layout and the
side effects of ``BA`` on ``alpha/beta/rs/cameras/spline/detections_global``.

Out of scope: trajectory initialisation, PnP-RANSAC, synchronisation search, plotting, and the
dead ``motion_prior=True`` branch -- those methods raise ``NotImplementedError``.  ``Scene.triangulate`` (SURVEY 8f rank
4) is here, with its per-point SVD on the GPU.

Extra ``settings`` keys (all optional; a reference ``config.json`` has none of them):

``ba_solver``  'trf' (DEFAULT) = scipy's TRF + LSMR restated on the GPU -- with ``ba_jacobian`` 'fd' (its default) this is
               the reference's algorithm end to end: same pattern matrix, same grouped 2-point differences, same trust
               region, same termination; results agree with the reference's within the reference's own reproducibility.
               'lm' = Levenberg-Marquardt on device-assembled normal equations with the Schur complement and the analytic
               Jacobian: ~20x faster per BA iteration, reaches a LOWER value of the same objective, hence another point than
               the reference's (DESIGN.md section 2 states how far, against the reference and against ground truth).
``ba_jacobian`` 'fd' (default with 'trf'), 'pattern' = analytic, masked to the reference sparsity pattern, 'analytic' =
               full analytic (default with 'lm').
``ba_pattern_ties`` 'numpy' = the twin rows of the pattern decided like np.argsort of this process decides them (default;
               what the reference would build here), 'canonical'.
``ba_lambda_min`` floor of the LM damping (default 3e-3, see ``mvus_solve_opts.lm_lambda_min``; the incremental loop of
               ``mvus_amd.pipeline`` sets 0.3 for its staged BAs unless told otherwise: ``LOOP_LM_LAMBDA_MIN`` below).
``ba_lm_wide_band`` with 'lm': what to do when the motion rows reach over more than six control points (knots less than a frame
               apart, i.e. more control points than detections): 'trf' (default) solves THAT problem with TRF + LSMR on the analytic
               Jacobian and says so, 'lm' keeps LM + Schur (general band solver, damping floor 0.3).
``ba_deterministic`` accepted and ignored: the 'lm' solver's normal equations are assembled without floating-point atomics
               (one writer, one order of additions per entry) -- the same bits on every run by construction.
``opt_sync`` (reference key: False freezes alpha/beta), ``device``.
"""
import json

import numpy as np

from ..tools import util
from .. import problem as _problem
from ..synth import rodrigues as _rodrigues, rotation_to_rvec as _rotation_to_rvec


def _undistort_normalized(points, K, d):
    """cv2.undistortPoints(src, K, d): 5 fixed-point iterations of the 5-coefficient model (host copy of
    ``undistort5`` in csrc/ba_math.h; used outside the BA loop only)."""
    k1, k2, p1, p2, k3 = [float(v) for v in np.asarray(d, dtype=np.float64).reshape(-1)[:5]]
    x0 = (points[0] - K[0, 2]) / K[0, 0]
    y0 = (points[1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    reset = np.zeros(np.shape(x), dtype=bool)        # OpenCV >= 4.1.1: a negative 1/(1 + k1 r^2 + ...) puts the point back at its start for good
    for _ in range(5):
        r2 = x * x + y * y
        with np.errstate(divide='ignore', invalid='ignore'):
            icd = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
        reset = reset | (icd < 0)
        dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
        dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
        x, y = np.where(reset, x0, (x0 - dx) * icd), np.where(reset, y0, (y0 - dy) * icd)
    return np.vstack((x, y))


# Floor of the LM damping that mvus_amd.pipeline.incremental_reconstruction sets (settings['ba_lambda_min']) when the loop runs with
# ba_solver = 'lm' and the caller gave none.  The staged BAs of the loop -- two or three cameras, similarity gauge free, a motion
# regulariser that is not scale invariant -- slope towards a smaller scene; a nearly undamped Newton step (the library's floor, 3e-3,
# chosen for single large BAs) follows that slope where the reference's ten truncated LSMR solves barely move.  Eleven seeds of the
# loop: with 0.3 every seed ends 0.21 - 0.43 m from the truth (the reference's algorithm: 0.23 - 0.51 m), with 3e-3 one seed in
# three ends metres away (profiles/round4/r04_loop_lm_damping_floors.txt).  A single Scene.BA keeps the library's floor: with 0.3 its last
# steps get short enough for xtol to end a 200-evaluation solve above the minimum (dist_fixed_2cam: cost 252.9 against 238.0).
LOOP_LM_LAMBDA_MIN = 0.3


class Camera:
    """K, R, t, d, P container (reference common.py:1040-1168)."""

    def __init__(self, **kwargs):
        self.P = kwargs.get('P')
        self.K = kwargs.get('K')
        self.R = kwargs.get('R')
        self.t = kwargs.get('t')
        self.d = kwargs.get('d')
        self.c = kwargs.get('c')
        self.fps = kwargs.get('fps')
        self.resolution = kwargs.get('resolution')

    def projectPoint(self, X):
        assert self.P is not None, 'The projection matrix P has not been calculated yet'
        if X.shape[0] == 3:
            X = util.homogeneous(X)
        x = self.P @ X
        return x / x[2]

    def compose(self):
        self.P = self.K @ np.hstack((self.R, np.asarray(self.t, dtype=np.float64).reshape(3, 1)))

    def decompose(self):
        M = self.P[:, :3]
        Rinv, Kinv = np.linalg.qr(np.linalg.inv(M))
        R, K = np.linalg.inv(Rinv), np.linalg.inv(Kinv)
        T = np.diag(np.sign(np.diag(K)))
        if np.linalg.det(T) < 0:
            T[1, 1] *= -1
        self.K, self.R = K @ T, T @ R
        self.t = np.linalg.inv(self.K) @ self.P[:, 3]
        self.K = self.K / self.K[-1, -1]
        return self.K, self.R, self.t

    def center(self):
        if self.c is None:
            self.decompose()
            self.c = -self.R.T @ self.t
        return self.c

    def P2vector(self, calib=False):
        r = _rotation_to_rvec(self.R)
        if calib:
            return np.concatenate(([self.K[0, 0], self.K[1, 1], self.K[0, 2], self.K[1, 2]], r, self.t, self.d))
        return np.concatenate((r, self.t))

    def vector2P(self, vector, calib=False):
        vector = np.asarray(vector, dtype=np.float64)
        if calib:
            self.K = np.diag((1.0, 1.0, 1.0))
            self.K[0, 0], self.K[1, 1] = vector[0], vector[1]
            self.K[:2, -1] = vector[2:4]
            self.R = _rodrigues(vector[4:7])
            self.t = vector[7:10].copy()
            self.d = vector[10:].copy()
        else:
            self.R = _rodrigues(vector[:3])
            self.t = vector[3:6].copy()
        self.compose()
        return self.P

    def undist_point(self, points):
        assert points.shape[0] == 2, 'Input must be a 2D array'
        return (self.K @ util.homogeneous(_undistort_normalized(points, self.K, self.d)))[:2]

    def info(self):
        for name in ('P', 'K', 'R', 't'):
            print('\n %s:' % name)
            print(getattr(self, name))


class Scene:
    """Everything known about the scene (reference common.py:22-61)."""

    def __init__(self):
        self.numCam = 0
        self.cameras = []
        self.detections = []
        self.detections_raw = []
        self.detections_global = []
        self.alpha = []
        self.beta = []
        self.beta_after_Fbeta = []
        self.cf = []
        self.traj = []
        self.traj_len = []
        self.sequence = []
        self.visible = []
        self.settings = []
        self.gt = []
        self.out = {}
        self.spline = {'tck': [], 'int': []}
        self.rs = []
        self.ref_cam = 0
        self.find_order = True
        self._ba_handle = None      # GPU handle kept between BA -> remove_outliers -> BA (main.py:49-62)
        self._ba_key = None

    def __getstate__(self):         # the output pickle (main.py:93) carries data only
        state = dict(self.__dict__)
        state['_ba_handle'] = None
        state['_ba_key'] = None
        return state

    # ---- bookkeeping --------------------------------------------------------------------------
    def addCamera(self, *camera):
        for c in camera:
            assert type(c) is Camera, 'camera is not an instance of Camera'
            self.cameras.append(c)

    def addDetection(self, *detection):
        for d in detection:
            assert d.shape[0] == 3, 'Detection must in form of (x,y,frameId)*N'
            self.detections.append(d)

    def init_alpha(self, *prior):
        if len(prior):
            assert len(prior) == self.numCam, 'Number of input must be the same as the number of cameras'
            self.alpha = prior
        else:
            fps_ref = self.cameras[self.ref_cam].fps
            self.alpha = np.array([fps_ref / self.cameras[i].fps for i in range(self.numCam)], dtype=np.float64)

    def time_shift(self, iter=False):
        assert len(self.cf) == self.numCam, 'The number of frame indices should equal to the number of cameras'
        if self.settings['cf_exact']:
            self.beta = self.cf[self.ref_cam] - self.alpha * self.cf
            print('The given corresponding frames are directly exploited as temporal synchronization\n')
        else:
            raise NotImplementedError('synchronisation search (sync_iter / sync_bf) is outside the BA hot path')

    def detection_to_global(self, *cam, motion_prior=False):
        """Frame indices -> global timeline, and the observed pixel (common.py:105-127)."""
        assert len(self.alpha) == self.numCam and len(self.beta) == self.numCam, 'The Number of alpha and beta is wrong'
        if motion_prior:
            raise NotImplementedError('motion_prior branch is dead in the reference pipeline')
        if len(cam):
            cams = cam if isinstance(cam[0], (int, np.integer)) else cam[0]
        else:
            cams = range(self.numCam)
            self.detections_global = [[] for _ in cams]
        for i in cams:
            det = self.detections[i]
            ts = self.alpha[i] * (det[0] + self.rs[i] * det[2] / self.cameras[i].resolution[1]) + self.beta[i]
            obs = self.cameras[i].undist_point(det[1:]) if self.settings['undist_points'] else det[1:]
            self.detections_global[i] = np.vstack((ts, obs))

    def cut_detection(self, second=1):
        if not second:
            return
        for i in range(self.numCam):
            det = self.detections[i]
            interval = util.find_intervals(det[0])
            cut = int(self.cameras[i].fps * second)
            keep = interval[:, interval[1] - interval[0] > cut * 2]
            keep[0] += cut
            keep[1] -= cut
            assert (keep[1] - keep[0] >= 0).all()
            self.detections[i], _ = util.sampling(det, keep)

    # ---- spline <-> trajectory: evaluation on the GPU, FITPACK's adaptive knot search on the host ----
    def traj_to_spline(self, smooth_factor):
        """One smoothing cubic spline per contiguous part of ``self.traj`` (reference common.py:224-270): FITPACK's ``splprep``
        inside the reference's knot-density loop, with the sample passes and banded solves on the GPU
        (``mvus_amd.spline.traj_fit`` -> ``mvus_spline_smooth``; no host fallback)."""
        from .. import spline as _spline
        assert len(smooth_factor) == 2, 'Smoothness should be defined by two parameters (min, max)'
        interval, idx = util.find_intervals(self.traj[0], idx=True)
        device = int(self.settings.get('device', 0)) if isinstance(self.settings, dict) else 0
        tck = [_spline.traj_fit(self.traj[:, idx[0, i]:idx[1, i] + 1], smooth_factor, device=device) for i in range(interval.shape[1])]
        self.spline['tck'], self.spline['int'] = tck, interval
        return self.spline

    def spline_to_traj(self, sampling_rate=1, t=None):
        """Discrete 3D points of the splines (common.py:273-301): sampled at a constant rate or at the given timestamps,
        kept where they lie inside an interval (closed ends), interval by interval.  The evaluation runs on the GPU
        (``mvus_spline_eval``, the FITPACK recurrence of the BA kernels) instead of scipy's splev."""
        from .. import spline as _spline
        tck, interval = self.spline['tck'], self.spline['int']
        if t is not None:
            assert len(t.shape) == 1, 'Input timestamps must be a 1D array'
            ts = np.asarray(t, dtype=np.float64)
        else:
            ts = np.arange(interval[0, 0], interval[1, -1], sampling_rate)
        device = int(self.settings.get('device', 0)) if isinstance(self.settings, dict) else 0
        X, which = _spline.evaluate(tck, interval, ts, device=device)
        parts = [np.empty([4, 0])]
        for i in range(interval.shape[1]):                       # the reference's order: interval by interval
            m = which == i
            if m.any():
                parts.append(np.vstack((ts[m], X[:, m])))
        self.traj = np.hstack(parts)
        assert (self.traj[0, 1:] >= self.traj[0, :-1]).all()
        return self.traj

    def all_detect_to_traj(self, *cam):
        """Attributes the reference refreshes while regularising (common.py:887-947); they do not enter the
        residual, so they are computed once after BA instead of on every evaluation."""
        cams = list(cam[0]) if len(cam) else list(range(self.numCam))
        for i in cams:
            self.detection_to_global(i)
        ts = np.concatenate([self.detections_global[i][0] for i in cams])
        frames = np.concatenate([self.detections[i][0] for i in cams])
        ids = np.concatenate([np.full(self.detections[i].shape[1], float(i)) for i in cams])
        self.frame_id_all = frames
        self.global_time_stamps_all = ts
        self.spline_to_traj(t=np.sort(ts))
        self.global_detections = np.vstack((ids, frames, ts))
        ordered = self.global_detections[:, np.argsort(ts)]
        ordered = ordered[:, np.isin(ordered[2], self.traj[0])]
        self.global_traj = np.vstack((np.arange(ordered.shape[1]), ordered, self.traj[1:]))
        assert (self.global_traj[3][1:] >= self.global_traj[3][:-1]).all(), 'timestamps are not in ascending order'

    # ---- the hot path --------------------------------------------------------------------------
    def _ba_problem(self, cams, rs=False, motion_reg=False, motion_weights=1, rs_bounds=False):
        st = self.settings
        return _problem.problem_from_arrays(
            [self.detections[i] for i in cams], [self.cameras[i] for i in cams], self.spline['tck'], self.spline['int'],
            opt_calib=st['opt_calib'], undist_points=st['undist_points'], rs=rs, rs_bounds=rs_bounds,
            motion_reg=motion_reg, motion_type=st.get('motion_type', 'F'), motion_weights=motion_weights,
            opt_sync=st.get('opt_sync', True))                      # absent -> alpha, beta free (common.py:512-515)

    def _pack(self, prob, cams):
        return _problem.pack_x(prob, np.asarray(self.alpha)[cams], np.asarray(self.beta)[cams], np.asarray(self.rs)[cams],
                               [self.cameras[i] for i in cams], self.spline['tck'])

    def _handle(self, prob):
        from ..ba import BAHandle          # raises if libmvusba.so is missing: no CPU fallback
        return BAHandle(prob, device=int(self.settings.get('device', 0)) if isinstance(self.settings, dict) else 0)

    @staticmethod
    def _same_problem(a, b):
        if (a.num_cam, a.opt_calib, a.undist_points, a.rs_free, a.rs_bounds, a.motion_reg, a.motion_type, a.motion_weight, a.opt_sync) != \
           (b.num_cam, b.opt_calib, b.undist_points, b.rs_free, b.rs_bounds, b.motion_reg, b.motion_type, b.motion_weight, b.opt_sync):
            return False
        pairs = [(a.det_offsets, b.det_offsets), (a.frame, b.frame), (a.u_raw, b.u_raw), (a.v_raw, b.v_raw),
                 (a.img_height, b.img_height), (a.interval, b.interval), (a.knot_offsets, b.knot_offsets), (a.knots, b.knots)]
        if not a.opt_calib:                      # with opt_calib K and d are parameters (in x), not problem data
            pairs += [(a.K, b.K), (a.dist, b.dist)]
        return all(x.shape == y.shape and np.array_equal(x, y) for x, y in pairs)

    def _resident_handle(self, prob, cams):
        """The handle of the previous BA over the same cameras if its (device-resident) problem is still this one."""
        h = self._ba_handle
        if h is not None and h.h and self._ba_key == tuple(cams) and self._same_problem(h.prob, prob):
            return h
        if h is not None:
            h.close()
        self._ba_handle, self._ba_key = self._handle(prob), tuple(cams)
        return self._ba_handle

    def _resident_for(self, cams):
        """The handle left by the last BA if it still describes the CURRENT scene state of its cameras (detections, knots,
        intervals, fixed calibration, flags) and covers ``cams``; else None.  x is packed from the current state."""
        h = self._ba_handle
        if h is None or not h.h or self._ba_key is None or not set(cams) <= set(self._ba_key):
            return None
        key = list(self._ba_key)
        if any(self.detections[i].shape[1] != int(h.prob.det_offsets[k + 1] - h.prob.det_offsets[k]) for k, i in enumerate(key)):
            return None
        now = self._ba_problem(key, rs=h.prob.rs_free, motion_reg=h.prob.motion_reg, motion_weights=h.prob.motion_weight,
                               rs_bounds=h.prob.rs_bounds)
        return h if self._same_problem(h.prob, now) else None

    def error_cam(self, cam_id, mode='dist', motion_prior=False, norm=False):
        """Reprojection errors of one camera (common.py:304-359), evaluated by the HIP residual kernel: on the handle the
        last BA left resident when it still describes the scene, else on a temporary residual-only handle."""
        if motion_prior or norm:
            raise NotImplementedError('motion_prior / norm variants are not on the BA hot path')
        self.detection_to_global(cam_id)
        h = self._resident_for([cam_id])
        if h is not None:
            key = list(self._ba_key)
            k = key.index(cam_id)
            a, b = int(h.prob.det_offsets[k]), int(h.prob.det_offsets[k + 1])
            f = h.residual(self._pack(h.prob, key))[2 * a:2 * b]
        else:
            prob = self._ba_problem([cam_id])
            with self._handle(prob) as tmp:
                f = tmp.residual(self._pack(prob, [cam_id]))
        M = self.detections[cam_id].shape[1]
        ex, ey = f[:M], f[M:2 * M]
        if mode == 'each':
            return np.concatenate((ex, ey))
        _, ids = util.sampling(self.detections_global[cam_id], self.spline['int'], belong=True)
        order = np.concatenate([np.nonzero(ids == s + 1)[0] for s in range(self.spline['int'].shape[1])]).astype(int)
        if mode == 'dist':
            return np.sqrt(ex[order] ** 2 + ey[order] ** 2)
        if mode == 'xy_1D':
            return np.concatenate((ex[order], ey[order]))
        if mode == 'xy_2D':
            return np.vstack((ex[order], ey[order]))
        raise ValueError('unknown mode %r' % (mode,))

    def compute_visibility(self):
        self.visible = []
        self.detection_to_global()
        for cam_id in range(self.numCam):
            _, vis = util.sampling(self.detections_global[cam_id], self.spline['int'], belong=True)
            self.visible.append(vis)

    @staticmethod
    def _motion_band_width(prob):
        """Control points a motion row couples (common.py:959-1001: samples j-1, j, j+1, four control points each), i.e. the block
        band width W of the spline part of J^T J; 4 without the regulariser."""
        if not prob.motion_reg:
            return 4
        ts, sid = prob.motion_sample_times()
        coff = prob.ctrl_offsets
        first = np.empty(ts.size, dtype=np.int64)
        for s in range(prob.S):
            t = prob.knots[int(prob.knot_offsets[s]):int(prob.knot_offsets[s + 1])]
            sel = sid == s
            first[sel] = coff[s] + np.clip(np.searchsorted(t, ts[sel], side='right') - 1, 3, t.size - 5) - 3
        W = 4
        step = 2 if prob.motion_type == 0 else 1            # 'F' rows use three samples, 'KE' two
        for d in range(1, step + 1):
            same = sid[d:] == sid[:-d]
            if same.any():
                W = max(W, int(np.max(np.abs(first[d:] - first[:-d])[same])) + 4)
        return W

    def BA(self, numCam, max_iter=10, rs=False, motion_prior=False, motion_reg=False, motion_weights=1, norm=False,
           rs_bounds=False, jac_sparsity=None):
        """Bundle adjustment over ``self.sequence[:numCam]`` (common.py:441-697): same arguments, same side effects,
        the optimisation itself runs in ``mvus_ba_solve`` on the GPU.  ``jac_sparsity`` (not in the reference's signature):
        the matrix ``jac_BA`` would return, for callers that have it -- the parity modes then use exactly that pattern
        instead of rebuilding it on the GPU (the reference hands the same matrix to ``least_squares``, common.py:670)."""
        if motion_prior:
            raise NotImplementedError('motion_prior=True is dead code in the reference pipeline (SURVEY.md section 2)')
        from .. import ba as _ba
        cams = list(self.sequence[:numCam])
        self.alpha, self.beta, self.rs = (np.asarray(v, dtype=np.float64) for v in (self.alpha, self.beta, self.rs))
        prob = self._ba_problem(cams, rs=rs, motion_reg=motion_reg, motion_weights=motion_weights, rs_bounds=rs_bounds)
        model = self._pack(prob, cams)
        print('Number of BA parameters is {}'.format(len(model)))
        print('Doing BA with {} cameras...\n'.format(numCam))
        st = self.settings
        solver, jac_mode = self.ba_mode()
        h = self._resident_handle(prob, cams)      # stays resident for remove_outliers and the next BA
        opts = _ba._lib.default_opts(solver, jac_mode, max_iter)
        opts.lm_lambda_min = float(st.get('ba_lambda_min', opts.lm_lambda_min))
        opts.lm_trust_radius = float(st.get('ba_trust_radius', opts.lm_trust_radius))
        try:
            if solver == _ba.SOLVER_LM_SCHUR and st.get('ba_lm_wide_band', 'trf') != 'lm' and self._motion_band_width(prob) > 6:
                # POLICY, not a limit of the library (it solves bands of up to sixteen control points, tests/test_gpu_schur.py): knots less
                # than a frame apart mean more control points than detections -- what traj_to_spline returns after a dense triangulate
                # inside the incremental loop -- and there an exact Newton-type step follows the regulariser-only directions: three
                # seeds of the loop end 2.6 - 8 m from the truth with LM on those problems, 0.2 - 0.35 m with the truncated solver
                # (profiles/round4/r04_loop_lm_wide_band.txt).  settings['ba_lm_wide_band'] = 'lm' keeps LM (damping floor 0.3).
                raise _ba.UnsupportedBySolver("settings['ba_lm_wide_band'] = 'trf': the motion rows reach over more than six control points (knots less than a frame apart)")
            res = h.solve(model, opts=opts, ties=st.get('ba_pattern_ties', 'numpy'), matrix=jac_sparsity)
            res.solver_used = 'lm' if solver == _ba.SOLVER_LM_SCHUR else 'trf'
        except _ba.UnsupportedBySolver as e:
            # LM + Schur keeps the spline block as a band of at most sixteen 3x3 blocks (MVUS_E_UNSUPPORTED beyond; FITPACK knots far
            # below one frame apart make the motion rows reach further).  The other GPU solver has no such limit: same analytic
            # Jacobian, TRF + LSMR instead of the normal equations.  Said aloud and recorded in the result, not silently.
            print('BA: %s -- solving this problem with ba_solver=trf (analytic Jacobian) instead' % e)
            opts = _ba._lib.default_opts(_ba.SOLVER_TRF_LSMR, _ba.JAC_ANALYTIC, max_iter)
            res = h.solve(model, opts=opts, ties='canonical')
            res.solver_used = 'trf (fallback from lm: %s)' % e
        alpha, beta, rs_new, cam_states, coefs = _problem.unpack_x(prob, res.x)
        self.alpha[cams], self.beta[cams], self.rs[cams] = alpha, beta, rs_new
        for k, i in enumerate(cams):
            c = self.cameras[i]
            if st['opt_calib']:
                c.K, c.d = cam_states[k]['K'], cam_states[k]['d']
            c.R, c.t = cam_states[k]['R'], cam_states[k]['t']
            c.compose()
        for s, c in enumerate(coefs):
            self.spline['tck'][s][1] = c
        self.detection_to_global()
        if motion_reg:
            self.all_detect_to_traj(cams)
            self.spline_to_traj()
        return res

    def ba_mode(self):
        """(solver, Jacobian mode) that ``BA`` runs with the current settings.  Without any ``ba_*`` key -- a reference
        config.json -- that is the reference's own algorithm: TRF + LSMR over grouped 2-point differences."""
        from .. import ba as _ba
        st = self.settings if isinstance(self.settings, dict) else {}
        name = st.get('ba_solver', 'trf')
        if name not in ('trf', 'lm'):
            raise ValueError("settings['ba_solver'] must be 'trf' or 'lm', not %r" % (name,))
        solver = _ba.SOLVER_LM_SCHUR if name == 'lm' else _ba.SOLVER_TRF_LSMR
        default_jac = 'analytic' if name == 'lm' else 'fd'
        modes = {'analytic': _ba.JAC_ANALYTIC, 'pattern': _ba.JAC_PATTERN, 'fd': _ba.JAC_FD}
        jac = st.get('ba_jacobian', default_jac)
        if jac not in modes:
            raise ValueError("settings['ba_jacobian'] must be one of %s, not %r" % (sorted(modes), jac))
        return solver, modes[jac]

    def remove_outliers(self, cams, thres=30, verbose=False):
        """Drop detections whose reprojection error is >= thres (common.py:700-717); the mask is computed by
        ``mvus_ba_outlier_mask``."""
        if not thres:
            return
        cams = list(cams)
        for i in cams:
            self.detection_to_global(i)
        # in place on the GPU only if the resident handle is over exactly these cameras AND still describes the current
        # scene (detections, knots, intervals, fixed K/d, undist_points): the reference always evaluates current state
        h = self._resident_for(cams) if self._ba_key == tuple(cams) else None
        if h is not None:
            # the detections of the last BA are still on the GPU: filter them there (mvus_ba_remove_outliers)
            prob_old = h.prob
            x = self._pack(prob_old, cams)
            f = h.residual(x) if verbose else None
            keep = h.remove_outliers(x, thres)
            prob = prob_old
        else:
            prob = self._ba_problem(cams)
            with self._handle(prob) as tmp:
                x = self._pack(prob, cams)
                keep = tmp.outlier_mask(x, thres)
                f = tmp.residual(x) if verbose else None
        for k, i in enumerate(cams):
            a, b = int(prob.det_offsets[k]), int(prob.det_offsets[k + 1])
            if verbose:
                ex, ey = f[2 * a:2 * a + (b - a)], f[2 * a + (b - a):2 * b]
                print('{} out of {} detections are removed for camera {}'.format(int((~keep[a:b]).sum()),
                                                                                  int(((ex != 0) | (ey != 0)).sum()), i))
            self.detections[i] = self.detections[i][:, keep[a:b]]
            self.detection_to_global(i)

    def triangulate(self, cam_id, cams, factor_t2s, factor_s2t=0.02, thres=0, refit=True, verbose=0):
        """Triangulate the detections of camera ``cam_id`` that lie outside the existing spline against the (already
        processed) cameras ``cams`` and append them to the trajectory (common.py:754-815).  Same steps as the reference;
        the per-point linear triangulation and both reprojection distances come from one GPU kernel
        (``mvus_triangulate``) instead of a Python loop of 4x4 SVDs."""
        from . import epipolar as ep
        new_cam = self.cameras[cam_id]
        assert new_cam.P is not None, 'The camera pose must be computed first'
        covered = self.spline['int']
        device = int(self.settings.get('device', 0)) if isinstance(self.settings, dict) else 0
        self.detection_to_global(cam_id)
        mine = self.detections_global[cam_id]
        mine = mine[:, ~np.asarray(util.sampling(mine, covered)[1], dtype=bool)]      # only what the spline does not cover yet

        def against(other):
            """[t; X] of this camera's uncovered detections paired in time with camera ``other`` (one GPU launch per pair of cameras)."""
            self.detection_to_global(other)
            try:
                a, b = util.match_overlap(mine, self.detections_global[other])
            except Exception:                       # no temporal overlap with this camera (the reference's bare except)
                return None
            Xh, d_new, d_other = ep.triangulate_with_errors(a[1:], b[1:], new_cam.P, self.cameras[other].P, device=device)
            pts = np.vstack((a[0], Xh[:3]))
            if thres:
                ok = (d_new < thres) & (d_other < thres)
                if verbose:
                    print('{} out of {} points are triangulated'.format(sum(ok), len(d_new)))
                pts = pts[:, ok]
            if verbose:
                print('{} points are triangulated into the 3D spline'.format(pts.shape[1]))
            return pts

        found = [pts for pts in map(against, cams) if pts is not None]
        X_new = np.hstack([np.empty([4, 0])] + found)
        assert not np.any(util.sampling(X_new, covered)[1]), 'Points should not be triangulated into the existing part of the 3D spline'
        self.spline_to_traj(sampling_rate=factor_s2t)
        merged = np.hstack((self.traj, X_new))
        self.traj = merged[:, np.unique(merged[0], return_index=True)[1]]      # time ordered, one sample per timestamp (the first)
        if refit:
            self.traj_to_spline(smooth_factor=factor_t2s)
        return X_new

    def get_camera_pose(self, cam_id, error=8, verbose=0):
        """Absolute pose of camera ``cam_id`` from the trajectory (reference common.py:719-750): the spline points at the
        camera's detection timestamps (GPU: ``mvus_spline_eval``) against the raw detections, PnP + RANSAC on the GPU
        (``mvus_pnp_ransac`` in place of ``cv2.solvePnPRansac``, reprojectionError = ``error``); sets R, t, P of the camera."""
        from .. import spline as _spline
        from .pnp import solve_pnp_ransac
        tck, interval = self.spline['tck'], self.spline['int']
        device = int(self.settings.get('device', 0)) if isinstance(self.settings, dict) else 0
        self.detection_to_global(cam_id)
        det = self.detections_global[cam_id]
        _, idx = util.sampling(det, interval, belong=True)
        detect, point_3D = np.empty([3, 0]), np.empty([3, 0])
        for i in range(interval.shape[1]):                       # the reference's order: interval by interval
            part = det[:, idx == i + 1]
            if part.size:
                X, which = _spline.evaluate([tck[i]], interval[:, i:i + 1], part[0], device=device)
                detect = np.hstack((detect, part))
                point_3D = np.hstack((point_3D, X))
        N = point_3D.shape[1]
        cam = self.cameras[cam_id]
        retval, rvec, tvec, inliers = solve_pnp_ransac(point_3D.T, detect[1:].T, cam.K, cam.d, reprojectionError=error, device=device)
        if not retval:
            raise ValueError('get_camera_pose: no pose is supported by six trajectory points within %g px' % error)
        cam.R = _rodrigues(np.ravel(rvec))
        cam.t = np.ravel(tvec)
        cam.compose()
        if verbose:
            print('{} out of {} points are inliers for PnP'.format(inliers.shape[0], N))

    def select_most_overlap(self, init=False):
        """The initial pair of cameras or the next camera with the largest temporal overlap (reference common.py:851-884): host
        bookkeeping over the detection timestamps; the resampled trajectory of the second branch comes from the GPU."""
        if not self.find_order:
            return
        self.detection_to_global()
        overlap_max = 0
        if init:
            init_pair = None
            for i in range(self.numCam - 1):
                for j in range(i + 1, self.numCam):
                    x, _ = util.match_overlap(self.detections_global[i], self.detections_global[j])
                    overlap = x.shape[1] / self.cameras[i].fps
                    if overlap > overlap_max:
                        overlap_max, init_pair = overlap, [i, j]
            self.sequence = init_pair
        else:
            traj = self.spline_to_traj()
            next_cam = None
            for i in [c for c in range(self.numCam) if self.cameras[c].P is None]:
                interval = util.find_intervals(self.detections_global[i][0])
                overlap = util.sampling(traj[0], interval)
                overlap = overlap[0] if isinstance(overlap, tuple) else overlap
                if len(overlap) > overlap_max:
                    overlap_max, next_cam = len(overlap), i
            self.sequence.append(next_cam)

    # ---- outside the hot path -----------------------------------------------------------------------
    def _out_of_scope(self, *a, **k):
        raise NotImplementedError('outside the BA hot path this package accelerates (SURVEY.md section 2); '
                                  'use the reference implementation for initialisation / synchronisation search')

    init_traj = plot_reprojection = _out_of_scope

    def error_motion(self, cams, mode='dist', norm=False, motion_weights=0, motion_reg=False, motion_prior=False):
        """The motion-regularisation rows of the BA residual (reference common.py:362-424 with ``motion_reg=True``): one value
        per sample of ``spline_to_traj()`` (unit steps of the reference camera's frame clock), 'F' or 'KE' by
        ``settings['motion_type']``, zero at the first/last samples of an interval.  Evaluated by ``k_motion`` on the GPU (the
        rows ``error_BA`` appends, common.py:462-467).  ``motion_prior=True`` is the reference's dead branch."""
        if motion_prior:
            raise NotImplementedError('motion_prior=True is dead code in the reference pipeline (SURVEY.md section 2)')
        if not motion_reg:
            raise ValueError('error_motion: motion_reg=True is the only live mode (the reference returns an unbound name otherwise)')
        cams = [int(cams)] if isinstance(cams, (int, np.integer)) else list(cams)
        for i in cams:
            self.detection_to_global(i)
        self.spline_to_traj()
        prob = self._ba_problem(cams[:1], motion_reg=True, motion_weights=motion_weights)
        with self._handle(prob) as tmp:
            f = tmp.residual(self._pack(prob, cams[:1]))
        out = f[2 * prob.M:]
        assert out.size == self.traj.shape[1], 'motion rows and trajectory samples do not correspond'
        return out


def create_scene(path_input):
    """Build a Scene from the reference's JSON config (common.py:1171-1230): sections
    'necessary inputs', 'optional inputs', 'settings'."""
    with open(path_input, 'r') as fh:
        config = json.load(fh)
    flight = Scene()
    flight.settings = config['settings']
    paths = config['necessary inputs']['path_detections']
    flight.numCam = len(paths)
    for p in paths:
        det = np.loadtxt(p, usecols=(2, 0, 1))[:flight.settings['num_detections']].T
        flight.addDetection(det)
    for p in config['necessary inputs']['path_cameras']:
        try:
            with open(p, 'r') as fh:
                cam = json.load(fh)
        except Exception:
            raise Exception('Wrong input of camera')
        d = list(cam['distCoeff'])
        if len(d) == 4:
            d.append(0)
        flight.addCamera(Camera(K=np.asarray(cam['K-matrix'], dtype=np.float64), d=np.asarray(d, dtype=np.float64),
                                fps=cam['fps'], resolution=cam['resolution']))
    flight.ref_cam = config['settings']['ref_cam']
    flight.sequence = config['settings']['camera_sequence']
    flight.find_order = False if len(flight.sequence) else True
    flight.cf = np.asarray(config['necessary inputs']['corresponding_frames'], dtype=np.float64)
    init_rs = config['settings']['init_rs'] if config['settings']['rolling_shutter'] else 0
    if isinstance(init_rs, list):
        assert len(init_rs) == flight.numCam, 'the number of initial rolling shutter values must equal the number of cameras'
        flight.rs = np.asarray(init_rs, dtype=np.float64)
    else:
        flight.rs = np.full(flight.numCam, float(init_rs))
    if 'optional inputs' in config and 'ground_truth' in config['optional inputs']:
        flight.gt = config['optional inputs']['ground_truth']
    print('Input data are loaded successfully, a scene is created.\n')
    return flight
