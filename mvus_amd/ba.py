"""Python face of the C ABI (include/mvus_ba.h): one ``BAHandle`` per BA problem, GPU and stream.

``BAHandle.solve`` stands where the reference calls ``scipy.optimize.least_squares``
(``reconstruction/common.py:670``); ``residual`` is ``error_BA`` (``common.py:448-487``);
``outlier_mask`` is the test of ``Scene.remove_outliers`` (``common.py:709-713``).
Everything is computed by libmvusba.so on the GPU; there is no CPU fallback.
"""
import ctypes
import sys
from types import SimpleNamespace

import numpy as np

from . import _lib
from ._lib import JAC_ANALYTIC, JAC_FD, JAC_PATTERN, SOLVER_LM_SCHUR, SOLVER_TRF_LSMR  # noqa: F401 (re-exported)

class UnsupportedBySolver(RuntimeError):
    """MVUS_E_UNSUPPORTED: the problem is outside what the chosen solver handles (see include/mvus_ba.h); the other solver has no such limit."""


class ReshardNeeded(RuntimeError):
    """MVUS_E_RESHARD: on a time shard the time stamps have drifted until a row reaches control points outside the slice.  ``x`` is the
    point the solver had reached, ``nfev`` the evaluations it used: re-cut there and continue (mvus_amd.dist.solve_time_sharded)."""
    x = None
    nfev = 0


_ERRORS = {_lib.MVUS_E_INVALID: ValueError, _lib.MVUS_E_NUMERIC: ValueError, _lib.MVUS_E_HIP: RuntimeError,
           _lib.MVUS_E_COMM: RuntimeError, _lib.MVUS_E_UNSUPPORTED: UnsupportedBySolver}


class _Result(SimpleNamespace):
    """OptimizeResult-like; ``active_mask`` (all zeros: the rs box is handled by projection) is built on first use."""

    @property
    def active_mask(self):
        return np.zeros(self.x.shape[0])


class BAHandle:
    def __init__(self, prob, device=0, stream=None):
        self.lib = _lib.load()
        self.prob = prob
        self._struct, self._keep = _lib.make_problem_struct(prob, device=device, stream=stream)
        h = ctypes.c_void_p()
        rc = self.lib.mvus_ba_create(ctypes.byref(self._struct), ctypes.byref(h))
        if rc != 0:
            raise _ERRORS.get(rc, RuntimeError)('mvus_ba_create: ' + self.lib.mvus_last_error(None).decode())
        self.h = h
        self.n = int(self.lib.mvus_ba_num_params(h))
        self.m = int(self.lib.mvus_ba_num_residuals(h))
        self.T = int(self.lib.mvus_ba_num_motion_rows(h))
        self.NS = int(self.lib.mvus_ba_num_slots(h))
        self.M = prob.M
        self._cb = None

    def close(self):
        if getattr(self, 'h', None):
            self.lib.mvus_ba_destroy(self.h)
            self.h = None

    def __del__(self):
        # not while the interpreter shuts down: modules (and torch's own HIP state) are torn down in no particular order then, and
        # a handle that dies with the process needs no hipFree -- one GPU-suite run in eleven ended with a fatal signal after its
        # summary line before this guard
        if sys is None or sys.is_finalizing():       # (module globals are already gone late in the shutdown)
            return
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise _ERRORS.get(rc, RuntimeError)('%s: %s' % (what, self.lib.mvus_last_error(self.h).decode()))

    @staticmethod
    def _x(x, n):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape != (n,):
            raise ValueError('x has shape %s, expected (%d,)' % (x.shape, n))
        return x

    def residual(self, x):
        x = self._x(x, self.n)
        f = np.empty(self.m)
        self._check(self.lib.mvus_ba_residual(self.h, _lib.dptr(x), _lib.dptr(f)), 'mvus_ba_residual')
        return f

    def residual_jacobian(self, x, jac_mode=JAC_ANALYTIC):
        """f[m], J[2, NS, M] (slot layout of include/mvus_ba.h), ctrl[M]."""
        x = self._x(x, self.n)
        f = np.empty(self.m)
        J = np.empty((2, self.NS, self.M))
        ctrl = np.empty(self.M, dtype=np.int32)
        self._check(self.lib.mvus_ba_residual_jacobian(self.h, _lib.dptr(x), jac_mode, _lib.dptr(f), _lib.dptr(J),
                                                       ctrl.ctypes.data_as(_lib.c_int32_p)), 'mvus_ba_residual_jacobian')
        return f, J, ctrl

    def motion_rows(self, x, jac_mode=JAC_ANALYTIC):
        x = self._x(x, self.n)
        mf = np.zeros(self.T)
        mJ = np.zeros((36, self.T))
        mctrl = np.full((3, self.T), -1, dtype=np.int32)
        self._check(self.lib.mvus_ba_motion_rows(self.h, _lib.dptr(x), jac_mode, _lib.dptr(mf), _lib.dptr(mJ),
                                                 mctrl.ctypes.data_as(_lib.c_int32_p)), 'mvus_ba_motion_rows')
        return mf, mJ, mctrl

    def set_pattern(self, x0, download=True):
        """Canonical pattern codes of the detection rows at x0, computed on the GPU (tie rows carry ``_lib.PAT_TIE``);
        ``download=False`` leaves them on the device only."""
        x0 = self._x(x0, self.n)
        pat = np.empty(self.M, dtype=np.int32) if download else None
        self._check(self.lib.mvus_ba_set_pattern(self.h, _lib.dptr(x0), pat.ctypes.data_as(_lib.c_int32_p) if download else None), 'mvus_ba_set_pattern')
        self._pattern_fixed = False      # (a pattern computed on the device is re-computed at the x0 of every JAC_PATTERN solve)
        return pat

    def motion_pattern(self):
        mp = np.empty(self.T, dtype=np.int32)
        self._check(self.lib.mvus_ba_motion_pattern(self.h, mp.ctypes.data_as(_lib.c_int32_p)), 'mvus_ba_motion_pattern')
        return mp

    def upload_pattern(self, pat, motion_pat=None):
        """Supply the pattern (codes of include/mvus_ba.h) instead of the GPU's canonical one -- the reference's matrix is
        an input of the call this library replaces (common.py:670)."""
        pat = np.ascontiguousarray(pat, dtype=np.int32)
        if pat.shape != (self.M,):
            raise ValueError('pat has shape %s, expected (%d,)' % (pat.shape, self.M))
        mp = None
        if motion_pat is not None and self.T:
            mp = np.ascontiguousarray(motion_pat, dtype=np.int32)
            if mp.shape != (self.T,):
                raise ValueError('motion_pat has shape %s, expected (%d,)' % (mp.shape, self.T))
        self._check(self.lib.mvus_ba_upload_pattern(self.h, pat.ctypes.data_as(_lib.c_int32_p),
                                                    mp.ctypes.data_as(_lib.c_int32_p) if mp is not None else None), 'mvus_ba_upload_pattern')
        self._pattern_fixed = True       # an uploaded pattern stays in force over every solve that follows (mvus_ba_solve keeps it)

    def set_deterministic(self, on=True):
        """Kept for the ABI: the LM + Schur assembly is window-major (one writer and one order per entry, no atomics) -- the same
        bits on every run whatever is set here."""
        self._check(self.lib.mvus_ba_set_deterministic(self.h, 1 if on else 0), 'mvus_ba_set_deterministic')

    def deterministic_fallback(self):
        """True when the last assembly went through the detection-major kernel with fp64 atomics (frames of a camera out of order,
        or normal equations formed from a stored non-analytic Jacobian): correct, not bit-reproducible."""
        v = ctypes.c_int32(0)
        self._check(self.lib.mvus_ba_deterministic_fallback(self.h, ctypes.byref(v)), 'mvus_ba_deterministic_fallback')
        return bool(v.value)

    def set_fd_groups(self, groups, num_groups):
        groups = np.ascontiguousarray(groups, dtype=np.int32)
        if groups.shape != (self.n,):
            raise ValueError('groups has shape %s, expected (%d,)' % (groups.shape, self.n))
        self._check(self.lib.mvus_ba_set_fd_groups(self.h, groups.ctypes.data_as(_lib.c_int32_p), int(num_groups)),
                    'mvus_ba_set_fd_groups')

    def prepare_pattern(self, x0, ties='numpy', matrix=None):
        """The reference pattern at x0 in force on the handle: canonical codes from the GPU, the twin rows decided
        by ``ties`` ('numpy': like np.argsort of this process, what the reference would build here; 'canonical': the
        library's choice), or -- ``matrix`` -- the codes a given reference matrix implies.  Returns (pat, motion_pat)."""
        from . import pattern
        if matrix is not None:
            pat, mpat = pattern.codes_from_matrix(self.prob, matrix)
        else:
            pat, mpat = pattern.resolve_ties(self.prob, x0, self.set_pattern(x0), self.motion_pattern() if self.T else None, how=ties)
        self.upload_pattern(pat, mpat if self.T else None)
        return pat, (mpat if self.T else None)

    def prepare_fd(self, x0, ties='numpy', matrix=None):
        """Set-up for JAC_FD at x0: the reference pattern (see prepare_pattern), scipy's column grouping on the host."""
        from . import pattern
        pat, mpat = self.prepare_pattern(x0, ties, matrix)
        groups, ng = pattern.fd_groups(self.prob, pat, mpat)
        self.set_fd_groups(groups, ng)
        return ng

    def jv(self, v):
        v = self._x(v, self.n)
        y = np.empty(self.m)
        self._check(self.lib.mvus_ba_jv(self.h, _lib.dptr(v), _lib.dptr(y)), 'mvus_ba_jv')
        return y

    def jtu(self, u):
        u = self._x(u, self.m)
        z = np.empty(self.n)
        self._check(self.lib.mvus_ba_jtu(self.h, _lib.dptr(u), _lib.dptr(z)), 'mvus_ba_jtu')
        return z

    def normal_equations(self):
        """g[n], JtJ_cam[C,B,B], band[N,W,3,3], cross[C,B,3N] of the Jacobian currently held."""
        C, B, N = self.prob.C, 3 + self.prob.P, int(self.prob.n_coef.sum())
        W = ctypes.c_int32(0)
        self._check(self.lib.mvus_ba_normal_equations(self.h, None, None, None, None, ctypes.byref(W)), 'mvus_ba_normal_equations')
        g = np.empty(self.n)
        cam = np.empty((C, B, B))
        band = np.empty((N, W.value, 3, 3))
        cross = np.empty((C, B, 3 * N))
        self._check(self.lib.mvus_ba_normal_equations(self.h, _lib.dptr(g), _lib.dptr(cam), _lib.dptr(band), _lib.dptr(cross),
                                                      ctypes.byref(W)), 'mvus_ba_normal_equations')
        return g, cam, band, cross

    def lm_step(self, lam):
        """p = -(J^T J + lam diag(J^T J))^-1 J^T f for the Jacobian / residual currently held (after residual_jacobian)."""
        p = np.empty(self.n)
        self._check(self.lib.mvus_ba_lm_step(self.h, float(lam), _lib.dptr(p)), 'mvus_ba_lm_step')
        return p

    def solve(self, x0, solver=SOLVER_TRF_LSMR, jac_mode=JAC_PATTERN, max_nfev=10, opts=None, return_fun=True,
              ties='numpy', matrix=None, prepared=False):
        """The least_squares call of Scene.BA.  Returns an OptimizeResult-like namespace
        (x, cost, fun, nfev, njev, status, optimality, ...).  ``ties`` / ``matrix``: how the reference pattern of the
        JAC_PATTERN / JAC_FD modes is fixed at x0 (see prepare_pattern); ``prepared``: the pattern (and column groups) UPLOADED by an
        earlier prepare_pattern / prepare_fd stay in force (a run of steps with ONE pattern, as least_squares keeps its
        jac_sparsity) -- a pattern that only exists on the device (set_pattern) would be recomputed at every call's x0, so that is
        refused here."""
        x = np.array(self._x(x0, self.n))
        o = opts if opts is not None else _lib.default_opts(solver, jac_mode, max_nfev)
        if prepared and o.jac_mode in (JAC_FD, JAC_PATTERN):
            if not getattr(self, '_pattern_fixed', False):
                raise ValueError('solve(prepared=True): call prepare_pattern / prepare_fd (or upload_pattern) first')
        elif o.jac_mode == JAC_FD:
            self.prepare_fd(x, ties, matrix)
        elif o.jac_mode == JAC_PATTERN and (matrix is not None or ties != 'canonical'):
            self.prepare_pattern(x, ties, matrix)
        elif o.jac_mode == JAC_PATTERN:
            self.set_pattern(x, download=False)      # canonical codes stay on the GPU: no host round trip (clears an uploaded pattern)
        res = _lib.MvusResult()
        f = np.empty(self.m) if return_fun else None
        rc = self.lib.mvus_ba_solve(self.h, _lib.dptr(x), ctypes.byref(o), ctypes.byref(res), _lib.dptr(f) if return_fun else None)
        if rc == _lib.MVUS_E_RESHARD:
            e = ReshardNeeded('mvus_ba_solve: %s' % self.lib.mvus_last_error(self.h).decode())
            e.x, e.nfev, e.cost, e.initial_cost = x, int(res.nfev), float(res.cost), float(res.initial_cost)
            raise e
        self._check(rc, 'mvus_ba_solve')
        return _Result(x=x, cost=res.cost, fun=f, nfev=res.nfev, njev=res.njev, status=res.status,
                       optimality=res.optimality, lin_iters=res.lin_iters, solve_ms=res.solve_ms,
                       initial_cost=res.initial_cost, success=res.status > 0, grad=None, jac=None)

    def outlier_mask(self, x, thres):
        """Per-detection ``error < thres`` (camera-segmented order), dtype bool."""
        x = self._x(x, self.n)
        keep = np.empty(self.M, dtype=np.uint8)
        self._check(self.lib.mvus_ba_outlier_mask(self.h, _lib.dptr(x), float(thres), keep.ctypes.data_as(_lib.c_uint8_p)),
                    'mvus_ba_outlier_mask')
        return keep.astype(bool)

    def remove_outliers(self, x, thres):
        """Scene.remove_outliers on the handle itself: the device-resident detections are filtered in place and the
        handle (and ``self.prob``) describe the inliers afterwards.  Returns the keep mask over the old detections."""
        x = self._x(x, self.n)
        keep = np.empty(self.M, dtype=np.uint8)
        off = np.zeros(self.prob.C + 1, dtype=np.int64)
        self._check(self.lib.mvus_ba_remove_outliers(self.h, _lib.dptr(x), float(thres), keep.ctypes.data_as(_lib.c_uint8_p),
                                                     off.ctypes.data_as(_lib.c_int64_p)), 'mvus_ba_remove_outliers')
        keep = keep.astype(bool)
        import dataclasses
        self.prob = dataclasses.replace(self.prob, det_offsets=off, frame=self.prob.frame[keep], u_raw=self.prob.u_raw[keep],
                                        v_raw=self.prob.v_raw[keep])
        self.M = self.prob.M
        self.m = int(self.lib.mvus_ba_num_residuals(self.h))
        assert self.m == self.prob.n_residuals
        return keep

    def set_x(self, x):
        x = self._x(x, self.n)
        self._check(self.lib.mvus_ba_set_x(self.h, _lib.dptr(x)), 'mvus_ba_set_x')

    def time_kernel(self, which, launches=20):
        """Average duration (ms) of ``launches`` back-to-back launches, HIP events on the handle's stream."""
        ms = ctypes.c_double(0.0)
        self._check(self.lib.mvus_ba_time_kernel(self.h, which, launches, ctypes.byref(ms)), 'mvus_ba_time_kernel')
        return ms.value

    def set_time_shard(self, rank, world, cuts, halo=8):
        """This handle holds time slice ``rank`` of ``world`` (BAProblem.shard_time): the LM/Schur solver keeps only the
        spline blocks of control points cuts[rank]..cuts[rank+1] (+- halo).  Call before set_allreduce."""
        cuts = np.ascontiguousarray(cuts, dtype=np.int32)
        assert cuts.size == world + 1
        self._check(self.lib.mvus_ba_set_time_shard(self.h, int(rank), int(world), cuts.ctypes.data_as(_lib.c_int32_p), int(halo)),
                    'mvus_ba_set_time_shard')

    def set_allreduce(self, fn, is_root=True):
        """fn(buf_dev_ptr:int, count:int, stream:int) -> None sums ``count`` doubles in place across ranks."""
        if fn is None:
            self._cb = _lib.ALLREDUCE_FN(0)
        else:
            def tramp(user, buf, count, stream):
                try:
                    fn(int(buf), int(count), int(stream or 0))
                    return 0
                except Exception:      # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = _lib.ALLREDUCE_FN(tramp)
        self._check(self.lib.mvus_ba_set_allreduce(self.h, self._cb, None, int(bool(is_root))), 'mvus_ba_set_allreduce')


    def set_rccl(self, unique_id, rank, world, is_root=True):
        """The sums over the ranks by RCCL called from the library (ncclAllReduce on the handle's stream; no Python in the iteration).
        ``unique_id``: the 128 bytes of _lib.rccl_unique_id() of ONE rank; collective -- returns once all ``world`` ranks have joined."""
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._cb = None
        self._check(self.lib.mvus_ba_set_rccl(self.h, buf, int(rank), int(world), int(bool(is_root))), 'mvus_ba_set_rccl')

    def time_allreduce(self, count, reps=50):
        """Average milliseconds per sum of ``count`` doubles through the installed route (callback or RCCL), HIP events on the handle's stream."""
        ms = ctypes.c_double(0.0)
        self._check(self.lib.mvus_ba_time_allreduce(self.h, int(count), int(reps), ctypes.byref(ms)), 'mvus_ba_time_allreduce')
        return ms.value


# kernel ids of mvus_ba_time_kernel
KERNEL_RESIDUAL, KERNEL_RESIDUAL_JACOBIAN, KERNEL_JV, KERNEL_JTU, KERNEL_ASSEMBLY, KERNEL_RESIDUAL_JACOBIAN_ONE_BUFFER, KERNEL_FUSED_ASSEMBLY = 0, 1, 2, 3, 4, 5, 6
