"""The reference's Jacobian sparsity pattern (``jac_BA``, reference common.py:490-610) and scipy's column
grouping for sparse finite differences, as host-side set-up for ``MVUS_JAC_FD``.

The detection rows come from the GPU (``mvus_ba_set_pattern``: per detection the first of the three in-pattern
control points, -1 for rows that are invisible at x0); motion rows are parameter independent and built here with the
same nearest-three rule (``pattern_first_kept`` in csrc/ba_math.h).  The grouping itself is scipy's
``group_columns`` (greedy colouring in the column order of ``RandomState(0).permutation(n)``,
scipy/optimize/_numdiff.py:216) -- the same call ``least_squares`` makes on the reference's matrix.
"""
import numpy as np

from . import bspline


def motion_pattern(prob):
    """Per motion row: first in-pattern control point (global index).  Rows whose sample is not a half-open member
    of any interval take the LAST spline, like the reference's ``tck[-1]`` (common.py:579-584)."""
    ts, sid = prob.motion_sample_times()
    iv = prob.interval
    coff = prob.ctrl_offsets
    out = np.zeros(ts.size, dtype=np.int64)
    member = np.zeros(ts.size, dtype=np.int64) - 1
    for s in range(prob.S):
        inside = (ts - iv[0, s] >= 0) != (ts - iv[1, s] >= 0)
        member[inside] = s
    use = np.where(member >= 0, member, prob.S - 1)
    for s in range(prob.S):
        m = use == s
        if not m.any():
            continue
        t = prob.knots[prob.knot_offsets[s]:prob.knot_offsets[s + 1]]
        tt = ts[m]
        l = bspline.find_span(t, tt)
        first = ((tt - t[l - 1]) > (t[l + 2] - tt)).astype(np.int64)
        out[m] = coff[s] + (l - 3) + first
    return out


def reference_pattern(prob, pat0):
    """scipy.sparse CSR matrix (m x n) of ones: the matrix ``jac_BA`` hands to ``least_squares``."""
    from scipy import sparse
    C, P, n = prob.C, prob.P, prob.n_params
    coff = prob.ctrl_offsets
    xoff = prob.spline_x_offsets
    ncoef = prob.n_coef
    pat0 = np.asarray(pat0, dtype=np.int64)

    def spline_cols(pc):
        s = np.searchsorted(coff, pc, side='right') - 1
        j = pc - coff[s]
        base = xoff[s] + j
        return np.stack([base + q + d * ncoef[s] for d in range(3) for q in range(3)], axis=1)

    rows, cols = [], []
    for c in range(C):
        a, b = int(prob.det_offsets[c]), int(prob.det_offsets[c + 1])
        vis = np.nonzero(pat0[a:b] >= 0)[0]
        cam_cols = [c, C + c] + ([2 * C + c] if prob.rs_free else []) + list(range(3 * C + c * P, 3 * C + (c + 1) * P))
        cc = np.concatenate((np.tile(np.array(cam_cols), (vis.size, 1)), spline_cols(pat0[a:b][vis])), axis=1)
        for r0 in (2 * a, 2 * a + (b - a)):
            rows.append(np.repeat(r0 + vis, cc.shape[1]))
            cols.append(cc.ravel())
    if prob.motion_reg:
        mp = motion_pattern(prob)
        cc = spline_cols(mp)
        rows.append(np.repeat(2 * prob.M + np.arange(mp.size), 9))
        cols.append(cc.ravel())
    rows = np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)
    cols = np.concatenate(cols) if cols else np.zeros(0, dtype=np.int64)
    A = sparse.csr_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)), shape=(prob.n_residuals, n))
    A.sum_duplicates()
    A.data[:] = 1
    return A


def fd_groups(prob, pat0):
    """Column groups for sparse 2-point differences: (groups int32[n], number of groups)."""
    from scipy.optimize._numdiff import group_columns
    groups = np.asarray(group_columns(reference_pattern(prob, pat0)), dtype=np.int32)
    return groups, int(groups.max()) + 1
