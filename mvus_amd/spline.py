"""Spline maintenance either side of BA on the GPU, through the C ABI:

``evaluate``   = the evaluation inside ``Scene.spline_to_traj`` (reference common.py:273-301; scipy ``splev``)
``lsq_fit``    = least-squares coefficients on a fixed knot vector (the refit step of ``traj_to_spline`` once the knots are
                 placed; scipy ``make_lsq_spline``)
``smooth_fit``  = ``scipy.interpolate.splprep(X, u=t, s=s, k=3)``: FITPACK's adaptive knot placement and smoothing-parameter search
                 with the sample passes and banded solves on the GPU (``mvus_spline_smooth``)
``traj_fit``    = the reference's smooth_factor loop around it (common.py:241-262): one interval of ``Scene.traj_to_spline``
No CPU fallback: without libmvusba.so / a GPU these raise."""
import ctypes

import numpy as np

from . import _lib


def _err(lib, rc, what):
    raise (ValueError if rc in (_lib.MVUS_E_INVALID, _lib.MVUS_E_NUMERIC) else RuntimeError)('%s: %s' % (what, lib.mvus_last_error(None).decode()))


def evaluate(tck, interval, t, device=0):
    """Points of the splines ``tck`` (list of [knots, [cx, cy, cz], 3]) at timestamps ``t`` (1-D).  Returns (X[3, len(t)],
    which[len(t)]): ``which`` is the interval each timestamp lies in (start <= t <= end) or -1, X is 0 where -1."""
    lib = _lib.load()
    interval = np.ascontiguousarray(np.asarray(interval, dtype=np.float64))
    S = interval.shape[1]
    knots = [np.asarray(k[0], dtype=np.float64) for k in tck]
    koff = np.concatenate(([0], np.cumsum([k.size for k in knots]))).astype(np.int64)
    coefs = np.concatenate([np.ravel(np.asarray(k[1], dtype=np.float64)) for k in tck])
    kn = np.ascontiguousarray(np.concatenate(knots))
    t = np.ascontiguousarray(t, dtype=np.float64)
    if t.ndim != 1:
        raise ValueError('Input timestamps must be a 1D array')
    X = np.zeros((3, t.size))
    which = np.full(t.size, -1, dtype=np.int32)
    rc = lib.mvus_spline_eval(int(device), int(S), _lib.dptr(interval), koff.ctypes.data_as(_lib.c_int64_p), _lib.dptr(kn), _lib.dptr(coefs),
                              t.size, _lib.dptr(t), _lib.dptr(X), which.ctypes.data_as(_lib.c_int32_p))
    if rc != 0:
        _err(lib, rc, 'mvus_spline_eval')
    return X, which


def lsq_fit(knots, t, X, device=0):
    """Coefficients [cx, cy, cz] of the least-squares cubic spline with knot vector ``knots`` through the points X[3, m] at t[m]."""
    lib = _lib.load()
    knots = np.ascontiguousarray(knots, dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    if X.shape != (3, t.size):
        raise ValueError('X must be 3 x len(t)')
    n = knots.size - 4
    c = np.empty((3, max(n, 1)))
    rc = lib.mvus_spline_lsq(int(device), int(knots.size), _lib.dptr(knots), t.size, _lib.dptr(t), _lib.dptr(X), _lib.dptr(c))
    if rc != 0:
        _err(lib, rc, 'mvus_spline_lsq')
    return [c[0].copy(), c[1].copy(), c[2].copy()]


def smooth_fit(t, X, s, device=0, full_output=False):
    """``scipy.interpolate.splprep(X, u=t, s=s, k=3)[0]``: the tck ``[knots, [cx, cy, cz], 3]`` of the smoothing cubic spline
    through X[3, m] at the strictly increasing timestamps t[m].  ``full_output``: also (fp, ier) as FITPACK reports them."""
    lib = _lib.load()
    t = np.ascontiguousarray(t, dtype=np.float64)
    X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    if t.ndim != 1 or X.shape != (3, t.size):
        raise ValueError('X must be 3 x len(t)')
    nest = t.size + 6
    n = ctypes.c_int32(0)
    ier = ctypes.c_int32(0)
    fp = ctypes.c_double(0.0)
    knots = np.zeros(nest)
    c = np.zeros((3, nest))
    rc = lib.mvus_spline_smooth(int(device), t.size, _lib.dptr(t), _lib.dptr(X), float(s), ctypes.byref(n), _lib.dptr(knots), _lib.dptr(c),
                                ctypes.byref(fp), ctypes.byref(ier))
    if rc != 0:
        _err(lib, rc, 'mvus_spline_smooth')
    nk = n.value
    tck = [knots[:nk].copy(), [c[d, :nk - 4].copy() for d in range(3)], 3]
    return (tck, fp.value, ier.value) if full_output else tck


class SmoothFit:
    """``smooth_fit`` for several smoothing factors on ONE set of samples: the samples are checked and uploaded once and the
    device work arrays are kept (``mvus_spline_fit_open`` / ``_smooth`` / ``_close``) -- what ``traj_fit``'s loop needs.

        with SmoothFit(t, X) as fit:
            tck = fit(s)
    """

    def __init__(self, t, X, device=0):
        self.lib = _lib.load()
        t = np.ascontiguousarray(t, dtype=np.float64)
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        if t.ndim != 1 or X.shape != (3, t.size):
            raise ValueError('X must be 3 x len(t)')
        self.m = t.size
        self.h = ctypes.c_void_p()
        rc = self.lib.mvus_spline_fit_open(int(device), t.size, _lib.dptr(t), _lib.dptr(X), ctypes.byref(self.h))
        if rc != 0:
            self.h = None
            _err(self.lib, rc, 'mvus_spline_fit_open')
        nest = self.m + 6
        self._knots = np.zeros(nest)
        self._c = np.zeros((3, nest))

    def __call__(self, s, full_output=False):
        if not self.h:
            raise ValueError('the fit session is closed')
        n = ctypes.c_int32(0)
        ier = ctypes.c_int32(0)
        fp = ctypes.c_double(0.0)
        rc = self.lib.mvus_spline_fit_smooth(self.h, float(s), ctypes.byref(n), _lib.dptr(self._knots), _lib.dptr(self._c), ctypes.byref(fp), ctypes.byref(ier))
        if rc != 0:
            _err(self.lib, rc, 'mvus_spline_fit_smooth')
        nk = n.value
        tck = [self._knots[:nk].copy(), [self._c[d, :nk - 4].copy() for d in range(3)], 3]
        return (tck, fp.value, ier.value) if full_output else tck

    def close(self):
        if getattr(self, 'h', None):
            self.lib.mvus_spline_fit_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:                                               # (at interpreter shutdown even `import sys` can fail)
            import sys
            if sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass


def _fprati(p1, f1, p2, f2, p3, f3):
    """FITPACK fprati: the zero of the rational interpolant r(p) = (u p + v) / (p + w) through three points (p3 < 0: p3 = inf)."""
    if p3 > 0.0:
        h1, h2, h3 = f1 * (f2 - f3), f2 * (f3 - f1), f3 * (f1 - f2)
        p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3)
    else:
        p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3)
    if f2 < 0.0:
        p3, f3 = p2, f2
    else:
        p1, f1 = p2, f2
    return p, p1, f1, p3, f3


def linear_fit_as_cubic(part, s):
    """The reference's fallback for a part with fewer than four samples (common.py:266-267): ``splprep(k=3)`` raises there
    (it needs m > k) and the bare ``except`` calls ``splprep(X, u=t, s=s, k=1)`` instead.  Through ``find_intervals`` only
    m = 3 can occur (two gaps < 5 spanning >= 5 time units).  This is FITPACK's ``fppara`` for k = 1 written out for that
    size (scipy 1.15.3, third party -- not under /root/reference; same flow, tolerances and iteration as the k = 3 case the
    GPU runs): the least-squares line if its squared residual is below s; otherwise one knot at the middle sample, and
    -- the interpolating polyline then has residual 0 < s -- the smoothing spline between the two with residual s, its
    parameter p found by fppara's own rational iteration (same starting value, brackets and stopping rule, so the same p).
    With the three linear B-splines being 1 at their own sample the observation matrix is the identity, which makes every
    step closed form: c(p) = x - b (b.x) / (p^2 + b.b), b = the derivative-jump row of ``fpdisc``.

    The kernels of this library evaluate CUBIC splines only, so the piecewise-linear curve is returned degree-elevated: the
    same curve, point for point, as a cubic tck (knots [a x4, b x4], or [a x4, u2 x3, b x4] for the bent one).  Documented
    divergence: ``tck[2]`` is 3 where the reference stores 1, and the knot / coefficient arrays differ accordingly; every
    evaluation (spline_to_traj, the BA residual) gives the reference's values."""
    part = np.asarray(part, dtype=np.float64)
    m = part.shape[1]
    u, X = part[0], part[1:]
    if m < 2:
        raise ValueError('traj_to_spline: a trajectory part with a single sample cannot be fitted (the reference raises too: splprep needs m > k)')
    a, b = u[0], u[-1]
    if m == 2:                                                        # nmax == nmin: the interpolating segment
        P, breaks = [X[:, 0], X[:, 1]], [a, b]
    elif m == 3:
        tol, maxit = 0.001, 20
        acc = tol * s
        lam = (u - a) / (b - a)
        A = np.column_stack((1.0 - lam, lam))
        c, *_ = np.linalg.lstsq(A, X.T, rcond=None)                  # (2, 3): values of the least-squares line at a and b
        fp0 = float(np.sum((A @ c - X.T) ** 2))
        if abs(fp0 - s) < acc or fp0 < s:                              # fppara: |fp - s| < acc, or fp < s with no interior knot (ier = -2)
            P, breaks = [c[0], c[1]], [a, b]
        else:
            fac = 2.0 / (b - a)                                        # fpdisc, k = 1, knots [a, a, u2, b, b]
            bj = np.array([1.0 / ((u[1] - a) * fac), (b - a) / ((u[1] - a) * (u[1] - b) * fac), 1.0 / ((b - u[1]) * fac)])
            bb, w = float(bj @ bj), X @ bj
            W = float(w @ w)
            F = lambda p: bb * W / (p * p + bb) ** 2                   # noqa: E731  residual of the smoothing spline with parameter p
            p1, f1, p3, f3 = 0.0, fp0 - s, -1.0, -s
            p = 1.0                                                    # nk1 / sum of the triangle's diagonal (three ones)
            ich1 = ich3 = 0
            for iteration in range(1, maxit + 1):
                f2 = F(p) - s
                if abs(f2) < acc or iteration == maxit:
                    break
                p2 = p
                if ich3 == 0:
                    if f2 - f3 <= acc:                                 # the initial choice of p is too large
                        p3, f3 = p2, f2
                        p = p * 0.04
                        if p <= p1:
                            p = p1 * 0.9 + p2 * 0.1
                        continue
                    if f2 < 0.0:
                        ich3 = 1
                if ich1 == 0:
                    if f1 - f2 <= acc:                                 # the initial choice of p is too small
                        p1, f1 = p2, f2
                        p = p / 0.04
                        if p3 >= 0.0 and p >= p3:
                            p = p2 * 0.1 + p3 * 0.9
                        continue
                    if f2 > 0.0:
                        ich1 = 1
                if f2 >= f1 or f2 <= f3:
                    break                                              # (fppara: ier = 2, keeps the current spline)
                p, p1, f1, p3, f3 = _fprati(p1, f1, p2, f2, p3, f3)
            Cp = X - np.outer(w, bj) / (p * p + bb)
            P, breaks = [Cp[:, 0], Cp[:, 1], Cp[:, 2]], [a, u[1], b]
    else:
        raise ValueError('linear_fit_as_cubic is the fallback for parts with fewer than four samples')
    knots = [breaks[0]] * 4
    coefs = [P[0]]
    for j in range(1, len(P)):
        d = P[j] - P[j - 1]
        coefs += [P[j - 1] + d / 3.0, P[j - 1] + 2.0 * d / 3.0, P[j]]
        knots += [breaks[j]] * (3 if j < len(P) - 1 else 4)
    C = np.array(coefs).T
    return [np.array(knots, dtype=np.float64), [C[0].copy(), C[1].copy(), C[2].copy()], 3]


def traj_fit(part, smooth_factor, device=0):
    """One interval of ``Scene.traj_to_spline`` (reference common.py:236-262): ``part`` = [t; x; y; z] (4, m).  The smoothing
    factor starts at 1e-6 * duration and is divided by 1.5 / doubled until duration / #coefficients lies between the two
    ``smooth_factor`` bounds (or the fit is down to 4 coefficients and still too dense).  Returns the tck."""
    lo, hi = min(smooth_factor), max(smooth_factor)
    part = np.ascontiguousarray(part, dtype=np.float64)     # once, not in each of the dozen fits below (a column slice of the trajectory is strided)
    measure = part[0, -1] - part[0, 0]
    s = (1e-3) ** 2 * measure
    if part.shape[1] < 4:                        # splprep(k=3) raises -> the reference's k=1 fallback (common.py:266-267)
        return linear_fit_as_cubic(part, s)
    prev, direction = 0, 0
    with SmoothFit(part[0], part[1:], device=device) as fit:       # one upload for the dozen fits of the loop
        while True:
            tck = fit(s)
            n = len(tck[0]) - 4
            if n == prev and n == 4 and direction == 2:
                break
            prev = n
            if measure / n > hi:
                s, direction = s / 1.5, 1
            elif measure / n < lo:
                s, direction = s * 2, 2
            else:
                break
    return tck
