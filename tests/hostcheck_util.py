"""Build/load the TEST-ONLY host build of the device math + optimiser (tests/hostcheck)."""
import ctypes
import os
import subprocess

import numpy as np

from mvus_amd import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, 'hostcheck', f) for f in ('hostcheck.cpp', 'host_backend.cpp')]
DEPS = SRC + [os.path.join(ROOT, 'mvus_amd', 'csrc', f) for f in ('ba_math.h', 'ba_solver.h', 'ba_problem.h', 'ba_schur.h', 'ba_partition.h', 'triangulate.hip.h', 'spline_fit.hip.h', 'pnp.hip.h')] \
    + [os.path.join(ROOT, 'include', 'mvus_ba.h')]
SO = os.path.join(HERE, 'hostcheck', 'libhostcheck.so')

_cached = None


def load():
    global _cached
    if _cached is not None:
        return _cached
    deps = [d for d in DEPS if os.path.exists(d)]
    if (not os.path.exists(SO)) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        flags = ['-DMVUS_WITH_SCHUR'] if os.path.exists(os.path.join(ROOT, 'mvus_amd', 'csrc', 'ba_schur.h')) else []
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC'] + flags + ['-o', SO] + SRC)
    lib = ctypes.CDLL(SO)
    lib.hostcheck_create.restype = ctypes.c_void_p
    lib.hostcheck_create.argtypes = [ctypes.POINTER(_lib.MvusProblem)]
    lib.hostcheck_destroy.argtypes = [ctypes.c_void_p]
    lib.hostcheck_error.restype = ctypes.c_char_p
    for name in ('hostcheck_n', 'hostcheck_m'):
        getattr(lib, name).restype = ctypes.c_int64
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    lib.hostcheck_T.argtypes = [ctypes.c_void_p]
    lib.hostcheck_residual.argtypes = [ctypes.c_void_p, _lib.c_double_p, _lib.c_double_p]
    lib.hostcheck_set_pattern.argtypes = [ctypes.c_void_p, _lib.c_double_p, _lib.c_int32_p]
    lib.hostcheck_dense_jacobian.argtypes = [ctypes.c_void_p, _lib.c_double_p, ctypes.c_int, _lib.c_double_p, _lib.c_double_p]
    lib.hostcheck_motion_pattern.argtypes = [ctypes.c_void_p, _lib.c_int32_p]
    lib.hostcheck_upload_pattern.argtypes = [ctypes.c_void_p, _lib.c_int32_p, _lib.c_int32_p]
    lib.hostcheck_set_fd_groups.argtypes = [ctypes.c_void_p, _lib.c_int32_p, ctypes.c_int]
    lib.hostcheck_jtu.argtypes = [ctypes.c_void_p, _lib.c_double_p, _lib.c_double_p]
    lib.hostcheck_solve.argtypes = [ctypes.c_void_p, _lib.c_double_p, ctypes.POINTER(_lib.MvusSolveOpts),
                                    ctypes.POINTER(_lib.MvusResult), _lib.c_double_p]
    lib.hostcheck_triangulate.argtypes = [ctypes.c_longlong] + [_lib.c_double_p] * 7
    lib.hostcheck_fpdisc.argtypes = [ctypes.c_int, _lib.c_double_p, _lib.c_double_p]
    lib.hostcheck_fprati.restype = ctypes.c_double
    lib.hostcheck_fprati.argtypes = [_lib.c_double_p]
    lib.hostcheck_fpknot.argtypes = [ctypes.c_int, _lib.c_double_p, _lib.c_int32_p, _lib.c_double_p, _lib.c_double_p, _lib.c_int32_p, _lib.c_int32_p]
    lib.hostcheck_fpknot_batch.argtypes = lib.hostcheck_fpknot.argtypes + [ctypes.c_int, ctypes.c_int]
    lib.hostcheck_dd.argtypes = [ctypes.c_int, _lib.c_double_p, _lib.c_double_p, _lib.c_double_p]
    lib.hostcheck_pnp_dlt6.argtypes = [_lib.c_double_p] * 4
    lib.hostcheck_pnp_project.argtypes = [_lib.c_double_p] * 6
    lib.hostcheck_pnp_point_normal.argtypes = [_lib.c_double_p] * 5 + [ctypes.c_double, ctypes.c_double, _lib.c_double_p]
    lib.hostcheck_pnp_sample6.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_longlong, _lib.c_int64_p]
    lib.hostcheck_rotation_to_rvec.argtypes = [_lib.c_double_p, _lib.c_double_p]
    _cached = lib
    return lib


class HostHandle:
    """Thin object wrapper over the hostcheck_* functions (mirrors mvus_amd.ba.BAHandle's methods)."""

    def __init__(self, prob):
        self.lib = load()
        self.prob = prob
        self._struct, self._keep = _lib.make_problem_struct(prob)
        self.h = self.lib.hostcheck_create(ctypes.byref(self._struct))
        if not self.h:
            raise ValueError(self.lib.hostcheck_error().decode())
        self.n = self.lib.hostcheck_n(self.h)
        self.m = self.lib.hostcheck_m(self.h)
        self.T = self.lib.hostcheck_T(self.h)

    def close(self):
        if self.h:
            self.lib.hostcheck_destroy(self.h)
            self.h = None

    def __del__(self):
        try:                                               # (at interpreter shutdown even `import sys` can fail)
            import sys
            if sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass

    def residual(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        f = np.zeros(self.m)
        self.lib.hostcheck_residual(self.h, _lib.dptr(x), _lib.dptr(f))
        return f

    def set_pattern(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        pat = np.zeros(self.prob.M, dtype=np.int32)
        self.lib.hostcheck_set_pattern(self.h, _lib.dptr(x0), pat.ctypes.data_as(_lib.c_int32_p))
        return pat

    def motion_pattern(self):
        mp = np.zeros(self.T, dtype=np.int32)
        self.lib.hostcheck_motion_pattern(self.h, mp.ctypes.data_as(_lib.c_int32_p))
        return mp

    def upload_pattern(self, pat, mpat=None):
        pat = np.ascontiguousarray(pat, dtype=np.int32)
        mp = np.ascontiguousarray(mpat, dtype=np.int32) if (mpat is not None and self.T) else None
        self.lib.hostcheck_upload_pattern(self.h, pat.ctypes.data_as(_lib.c_int32_p), mp.ctypes.data_as(_lib.c_int32_p) if mp is not None else None)

    def prepare_pattern(self, x0, ties='numpy', matrix=None):
        from mvus_amd import pattern
        if matrix is not None:
            pat, mpat = pattern.codes_from_matrix(self.prob, matrix)
        else:
            pat, mpat = pattern.resolve_ties(self.prob, x0, self.set_pattern(x0), self.motion_pattern() if self.T else None, how=ties)
        self.upload_pattern(pat, mpat)
        return pat, (mpat if self.T else None)

    def prepare_fd(self, x0, ties='numpy', matrix=None):
        from mvus_amd import pattern
        pat, mpat = self.prepare_pattern(x0, ties, matrix)
        groups, ng = pattern.fd_groups(self.prob, pat, mpat)
        self._groups = np.ascontiguousarray(groups, dtype=np.int32)
        self.lib.hostcheck_set_fd_groups(self.h, self._groups.ctypes.data_as(_lib.c_int32_p), ng)
        return pat, self._groups, ng

    def dense_jacobian(self, x, jac_mode):
        x = np.ascontiguousarray(x, dtype=np.float64)
        f = np.zeros(self.m)
        J = np.zeros((self.m, self.n))
        self.lib.hostcheck_dense_jacobian(self.h, _lib.dptr(x), jac_mode, _lib.dptr(f), _lib.dptr(J))
        return f, J

    def jtu(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        z = np.zeros(self.n)
        self.lib.hostcheck_jtu(self.h, _lib.dptr(u), _lib.dptr(z))
        return z

    def solve(self, x0, opts, ties='numpy', matrix=None):
        x = np.array(x0, dtype=np.float64)
        if opts.jac_mode == _lib.JAC_FD:
            self.prepare_fd(x, ties, matrix)
        elif opts.jac_mode == _lib.JAC_PATTERN:
            self.prepare_pattern(x, ties, matrix)
        res = _lib.MvusResult()
        f = np.zeros(self.m)
        rc = self.lib.hostcheck_solve(self.h, _lib.dptr(x), ctypes.byref(opts), ctypes.byref(res), _lib.dptr(f))
        if rc != 0:
            raise ValueError(self.lib.hostcheck_error().decode())
        return x, res, f


def host_triangulate(x1, x2, P1, P2, errors=True):
    """Host build of mvus_amd/csrc/triangulate.hip.h: X[4,N] (and the two reprojection distances)."""
    lib = load()
    x1, x2 = np.ascontiguousarray(x1[:2], dtype=np.float64), np.ascontiguousarray(x2[:2], dtype=np.float64)
    P1, P2 = np.ascontiguousarray(P1, dtype=np.float64), np.ascontiguousarray(P2, dtype=np.float64)
    N = x1.shape[1]
    X, e1, e2 = np.zeros((4, N)), np.zeros(N), np.zeros(N)
    lib.hostcheck_triangulate(N, _lib.dptr(x1), _lib.dptr(x2), _lib.dptr(P1), _lib.dptr(P2), _lib.dptr(X), _lib.dptr(e1), _lib.dptr(e2))
    return (X, e1, e2) if errors else X
