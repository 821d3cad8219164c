"""Host-side cubic B-spline helpers (numpy).

Used by the synthetic-scene generator and by the host mirror of the reference
``Scene`` for set-up work *outside* the BA hot path (the hot path evaluates
splines on the GPU, see ``csrc/ba_math.h``).  The spline representation is the
FITPACK one the reference stores in ``Scene.spline['tck']``
(reference ``reconstruction/common.py:224-270``): knot vector ``t`` of length
``n+4`` whose first and last four knots are equal, ``n`` coefficients per axis.
"""
import numpy as np


def find_span(t, x):
    """Index l with t[l] <= x < t[l+1], clamped to [3, n-1] (n = len(t)-4)."""
    t = np.asarray(t, dtype=np.float64)
    n = t.size - 4
    l = np.searchsorted(t, x, side='right') - 1
    return np.clip(l, 3, n - 1)


def basis(t, x, l=None):
    """Values of the 4 cubic B-splines that are non-zero on span l, shape (4, k)."""
    t = np.asarray(t, dtype=np.float64)
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    if l is None:
        l = find_span(t, x)
    h = np.zeros((4, x.size))
    h[0] = 1.0
    for j in range(1, 4):
        hh = h.copy()
        h[:] = 0.0
        for i in range(j):
            li = l + i + 1
            lj = li - j
            f = hh[i] / (t[li] - t[lj])
            h[i] += f * (t[li] - x)
            h[i + 1] = f * (x - t[lj])
    return h


def evaluate(t, c, x):
    """Spline value at x; c has shape (dim, n); returns (dim, k)."""
    c = np.asarray(c, dtype=np.float64)
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    l = find_span(t, x)
    h = basis(t, x, l)
    out = np.zeros((c.shape[0], x.size))
    for q in range(4):
        out += c[:, l - 3 + q] * h[q]
    return out


def make_knots(start, end, spacing, rng=None, jitter=0.25):
    """Clamped cubic knot vector on [start, end] with ~spacing between interior knots.

    A little jitter makes the knots non-uniform, like the FITPACK-placed knots the
    reference gets from ``splprep`` (reference ``common.py:247``)."""
    n_int = max(int(round((end - start) / spacing)) - 1, 0)
    interior = start + (end - start) * (np.arange(1, n_int + 1) / (n_int + 1))
    if rng is not None and n_int:
        h = (end - start) / (n_int + 1)
        interior = interior + rng.uniform(-jitter, jitter, n_int) * h
    return np.concatenate(([start] * 4, interior, [end] * 4)).astype(np.float64)


def lsq_fit(t, x, y):
    """Least-squares coefficients (dim, n) of the spline with knots t through (x, y)."""
    t = np.asarray(t, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = t.size - 4
    l = find_span(t, x)
    h = basis(t, x, l)
    # banded normal equations assembled densely in blocks is fine for set-up sizes,
    # but keep memory O(k*4): build (AtA) as a dense n x n only when n is small.
    if n <= 4000:
        A = np.zeros((x.size, n))
        rows = np.arange(x.size)
        for q in range(4):
            A[rows, l - 3 + q] = h[q]
        c, *_ = np.linalg.lstsq(A, y.T, rcond=None)
        return c.T
    # large n: banded normal equations via scipy
    from scipy.linalg import solveh_banded
    ab = np.zeros((4, n))
    rhs = np.zeros((n, y.shape[0]))
    for q in range(4):
        np.add.at(rhs, l - 3 + q, (h[q] * y).T)
        for p in range(q, 4):
            np.add.at(ab[3 - (p - q)], l - 3 + p, h[q] * h[p])
    ab[3] += 1e-12
    return solveh_banded(ab, rhs).T
