"""Host-side interval helpers with the reference's names and semantics (``tools/util.py`` there).

Only what the BA path and its immediate callers need: ``homogeneous`` (util.py:54), ``find_intervals``
(util.py:58-87), ``sampling`` (util.py:90-116), ``match_overlap`` (util.py:119-135, for the ground-truth report).
Written against the behaviour, vectorised over intervals.
"""
import numpy as np


def homogeneous(x):
    """Append a row of ones."""
    x = np.asarray(x)
    return np.concatenate((x, np.ones((1, x.shape[1]), dtype=x.dtype)), axis=0)


def find_intervals(x, gap=5, idx=False):
    """Start/end values of the maximal runs of ``x`` whose consecutive spacing is < ``gap``; runs shorter
    than ``gap`` are dropped.  ``x`` must be strictly ascending.  With ``idx`` also the index pairs."""
    x = np.asarray(x)
    assert x.ndim == 1 and np.all(np.diff(x) > 0), 'Input must be an ascending 1D-array'
    if x.size == 0:
        empty = np.zeros((2, 0))
        return (empty, empty.astype(int)) if idx else empty
    breaks = np.nonzero(np.diff(x) >= gap)[0]
    first = np.concatenate(([0], breaks + 1))
    last = np.concatenate((breaks, [x.size - 1]))
    long_enough = x[last] - x[first] >= gap
    first, last = first[long_enough], last[long_enough]
    interval = np.array([x[first], x[last]])
    assert np.all(interval[0, 1:] > interval[1, :-1])
    if idx:
        return interval, np.array([first, last])
    return interval


def interval_index(timestamp, interval):
    """1-based index of the half-open interval [start, end) containing each timestamp, 0 for none."""
    timestamp = np.asarray(timestamp, dtype=np.float64)
    out = np.zeros(timestamp.shape, dtype=int)
    for i in range(interval.shape[1]):                         # later intervals win, as in the reference loop
        inside = (timestamp - interval[0, i] >= 0) != (timestamp - interval[1, i] >= 0)
        out[inside] = i + 1
    return out


def sampling(x, interval, belong=False):
    """Points of ``x`` (1-D timestamps, or 3/4-row array with timestamps in row 0) that fall inside the
    intervals, plus the membership (bool mask, or 1-based interval ids with ``belong``)."""
    x = np.asarray(x)
    if x.ndim == 1:
        ts = x
    elif x.ndim == 2:
        assert x.shape[0] in (3, 4), 'Input should be 1D array or 2D array with 3 or 4 rows'
        ts = x[0]
    else:
        raise Exception('The shape of input is wrong')
    ids = interval_index(ts, interval)
    mask = ids.astype(bool)
    picked = x[mask] if x.ndim == 1 else x[:, mask]
    return picked, (ids if belong else mask)


def match_overlap(x, y):
    """The parts of two [t; x; y; z] tracks on one timeline that overlap in time: ``x`` restricted to the contiguous
    parts of ``y``, and ``y`` interpolated (cubic, through its samples) at those timestamps.  ``x`` is assumed to be the
    denser one."""
    from scipy import interpolate
    x_s, _ = sampling(x, find_intervals(y[0]))
    tck, _ = interpolate.splprep(y[1:], u=y[0], s=0, k=3)
    y_s = np.vstack((x_s[0], np.asarray(interpolate.splev(x_s[0], tck))))
    assert (x_s[0] == y_s[0]).all(), 'Both outputs should have the same timestamps'
    return x_s, y_s
