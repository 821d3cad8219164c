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


def traj_fit(part, smooth_factor, device=0):
    """One interval of ``Scene.traj_to_spline`` (reference common.py:236-262): ``part`` = [t; x; y; z] (4, m).  The smoothing
    factor starts at 1e-6 * duration and is divided by 1.5 / doubled until duration / #coefficients lies between the two
    ``smooth_factor`` bounds (or the fit is down to 4 coefficients and still too dense).  Returns the tck."""
    lo, hi = min(smooth_factor), max(smooth_factor)
    measure = part[0, -1] - part[0, 0]
    s = (1e-3) ** 2 * measure
    prev, direction = 0, 0
    while True:
        tck = smooth_fit(part[0], part[1:], s, device=device)
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
