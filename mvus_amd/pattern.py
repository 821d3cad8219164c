"""The reference's Jacobian sparsity pattern (``jac_BA``, reference common.py:490-610) and scipy's column
grouping for sparse finite differences, as host-side set-up for ``MVUS_JAC_PATTERN`` / ``MVUS_JAC_FD``.

The library describes the spline part of a row by a *pattern code* ``p | (mask << 25)`` (include/mvus_ba.h): ``p`` the
global index of the lowest in-pattern control point, bit ``k`` of the 4-bit mask set when ``p + k`` is in the pattern.
Detection-row codes come from the GPU (``mvus_ba_set_pattern``), motion-row codes from ``mvus_ba_motion_pattern``;
both are *canonical*: three consecutive control points (mask 7), the nearest three centre knots of ``t[2:-2]``.

``t[2:-2]`` repeats the interval start and end (coefficients 0,1 and n-2,n-1 share a centre knot).  When exactly one
member of such a pair is among the three nearest, which one ``np.argsort`` (common.py:561,582, default unstable kind)
returns is decided by numpy's sort kernel -- the scalar, AVX2 and AVX512 builds of numpy 2.2 give different answers on
the same input -- so in those rows (flagged ``PAT_TIE`` by the library) the reference's own matrix is implementation
defined.  ``resolve_ties(..., how='numpy')`` asks the numpy of *this* process the same question the reference asks
(an integer decision about ties, no arithmetic of the path), so that ``Scene.BA`` in its parity modes hands the
optimiser the matrix the reference would have built on this machine; ``codes_from_matrix`` converts a matrix that
is already at hand (the reference passes it into ``least_squares`` as an input).

The grouping is scipy's ``group_columns`` (greedy colouring in the column order of
``RandomState(0).permutation(n)``, scipy/optimize/_numdiff.py:216) -- the same call ``least_squares`` makes on the
reference's matrix.
"""
import ctypes

import numpy as np

from ._lib import PAT_SHIFT, PAT_TIE

PAT_INDEX_MASK = (1 << PAT_SHIFT) - 1
PAT_CANON = 7 << PAT_SHIFT


def code_index(code):
    return np.asarray(code, dtype=np.int64) & PAT_INDEX_MASK


def code_mask(code):
    return (np.asarray(code, dtype=np.int64) >> PAT_SHIFT) & 0xF


def is_tie(code):
    code = np.asarray(code, dtype=np.int64)
    return (code >= 0) & ((code & PAT_TIE) != 0)


def strip_flags(code):
    """Codes without the output-only tie flag."""
    code = np.asarray(code, dtype=np.int64)
    return np.where(code >= 0, code & ~np.int64(PAT_TIE), code).astype(np.int32)


def _code_from_points(points):
    """Pattern code of a sorted triple of global control-point indices, or -1 if they do not fit 4 consecutive points."""
    p = int(points[0])
    mask = 0
    for g in points:
        d = int(g) - p
        if d > 3:
            return -1
        mask |= 1 << d
    return p | (mask << PAT_SHIFT)


def detection_timestamps(prob, x0):
    """tau at x0 exactly as the reference forms it (detection_to_global, common.py:125)."""
    C = prob.C
    alpha, beta, rs = x0[:C], x0[C:2 * C], x0[2 * C:3 * C]
    tau = np.empty(prob.M)
    for c in range(C):
        a, b = int(prob.det_offsets[c]), int(prob.det_offsets[c + 1])
        tau[a:b] = alpha[c] * (prob.frame[a:b] + rs[c] * prob.v_raw[a:b] / prob.img_height[c]) + beta[c]
    return tau


def _resolve(prob, codes, times, how):
    codes = np.asarray(codes, dtype=np.int64).copy()
    rows = np.nonzero(is_tie(codes))[0]
    out = strip_flags(codes).astype(np.int64)
    if how == 'canonical' or rows.size == 0:
        return out.astype(np.int32)
    if how != 'numpy':
        raise ValueError("how must be 'numpy' or 'canonical'")
    coff = prob.ctrl_offsets
    for r in rows:
        p = int(codes[r] & PAT_INDEX_MASK)
        s = int(np.searchsorted(coff, p, side='right') - 1)
        t = prob.knots[int(prob.knot_offsets[s]):int(prob.knot_offsets[s + 1])]
        knot = t[2:-2]
        idx = np.sort(np.argsort(abs(knot - times[r]))[:3]) + int(coff[s])      # the reference's own question, common.py:561
        c = _code_from_points(idx)
        if c >= 0:
            out[r] = c
    return out.astype(np.int32)


def resolve_ties(prob, x0, pat, motion_pat=None, how='numpy'):
    """(pat, motion_pat) with the flagged twin rows decided: 'numpy' = like np.argsort of this process,
    'canonical' = keep the library's choice.  The tie flag is cleared either way."""
    pat = _resolve(prob, pat, detection_timestamps(prob, np.asarray(x0, dtype=np.float64)), how)
    if motion_pat is not None and len(motion_pat):
        ts, _ = prob.motion_sample_times()
        motion_pat = _resolve(prob, motion_pat, ts, how)
    return pat, motion_pat


def codes_from_matrix(prob, A):
    """Pattern codes (pat[M], motion_pat[T]) implied by a reference matrix ``A`` (m x n, scipy sparse or dense;
    the ``jac_sparsity`` argument of common.py:670): the spline columns of the x row of every detection, and of every
    motion row.  Rows without spline columns give -1."""
    from scipy import sparse
    A = sparse.csr_matrix(A)
    C, P = prob.C, prob.P
    first_spline_col = C * (3 + P)
    xoff, ncoef, coff = prob.spline_x_offsets, prob.n_coef, prob.ctrl_offsets

    def row_code(r):
        cols = A.indices[A.indptr[r]:A.indptr[r + 1]]
        cols = np.sort(cols[cols >= first_spline_col])
        if cols.size == 0:
            return -1
        s = int(np.searchsorted(xoff, cols[0], side='right') - 1)
        local = cols[cols < xoff[s] + ncoef[s]] - xoff[s]            # the x-coordinate block of spline s
        if local.size != 3 or cols.size != 9:
            raise ValueError('row %d of the matrix is not a three-control-point pattern' % r)
        c = _code_from_points(local + int(coff[s]))
        if c < 0:
            raise ValueError('row %d: the pattern does not fit four consecutive control points' % r)
        return c

    pat = np.full(prob.M, -1, dtype=np.int32)
    for c in range(C):
        a, b = int(prob.det_offsets[c]), int(prob.det_offsets[c + 1])
        for i in range(a, b):
            pat[i] = row_code(2 * a + (i - a))
    T = prob.num_motion_rows
    mpat = np.array([row_code(2 * prob.M + j) for j in range(T)], dtype=np.int32)
    return pat, mpat


def pattern_entries(prob, pat0, motion_pat=None):
    """(rows, cols) int64 of the ones of ``jac_BA``'s matrix (unordered; a (row, col) pair can occur twice), from pattern codes."""
    C, P, n = prob.C, prob.P, prob.n_params
    coff = prob.ctrl_offsets
    xoff = prob.spline_x_offsets
    ncoef = prob.n_coef
    pat0 = np.asarray(pat0, dtype=np.int64)

    def spline_entries(codes, row_ids):
        """(rows, cols) of the spline columns of the rows with these codes."""
        p = codes & PAT_INDEX_MASK
        mk = (codes >> PAT_SHIFT) & 0xF
        s = np.searchsorted(coff, p, side='right') - 1
        base = xoff[s] + (p - coff[s])
        rr, cc = [], []
        for k in range(4):
            sel = ((mk >> k) & 1) == 1
            for d in range(3):
                rr.append(row_ids[sel])
                cc.append(base[sel] + k + d * ncoef[s][sel])
        return np.concatenate(rr), np.concatenate(cc)

    rows, cols = [], []
    for c in range(C):
        a, b = int(prob.det_offsets[c]), int(prob.det_offsets[c + 1])
        vis = np.nonzero(pat0[a:b] >= 0)[0]
        cam_cols = ([c, C + c] if getattr(prob, 'opt_sync', True) else []) + ([2 * C + c] if prob.rs_free else []) \
            + list(range(3 * C + c * P, 3 * C + (c + 1) * P))
        sr, sc = spline_entries(pat0[a:b][vis], vis)
        for r0 in (2 * a, 2 * a + (b - a)):
            rows.append(np.repeat(r0 + vis, len(cam_cols)))
            cols.append(np.tile(np.array(cam_cols, dtype=np.int64), vis.size))
            rows.append(r0 + sr)
            cols.append(sc)
    if prob.motion_reg:
        if motion_pat is None:
            raise ValueError('motion_reg: pass the motion-row codes (mvus_ba_motion_pattern)')
        mp = np.asarray(motion_pat, dtype=np.int64)
        sr, sc = spline_entries(mp, np.arange(mp.size))
        rows.append(2 * prob.M + sr)
        cols.append(sc)
    rows = np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)
    cols = np.concatenate(cols) if cols else np.zeros(0, dtype=np.int64)
    return np.ascontiguousarray(rows, dtype=np.int64), np.ascontiguousarray(cols, dtype=np.int64)


def reference_pattern(prob, pat0, motion_pat=None):
    """scipy.sparse CSR matrix (m x n) of ones: the matrix ``jac_BA`` hands to ``least_squares``, from pattern codes."""
    from scipy import sparse
    rows, cols = pattern_entries(prob, pat0, motion_pat)
    A = sparse.csr_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)), shape=(prob.n_residuals, prob.n_params))
    A.sum_duplicates()
    A.data[:] = 1
    return A


def fd_groups_scipy(prob, pat0, motion_pat=None):
    """``scipy.optimize._numdiff.group_columns`` itself on the reference pattern -- what ``least_squares`` does with
    ``jac_sparsity`` (reference common.py:665-670); the check of ``fd_groups``."""
    from scipy.optimize._numdiff import group_columns
    groups = np.asarray(group_columns(reference_pattern(prob, pat0, motion_pat)), dtype=np.int32)
    return groups, int(groups.max()) + 1


def fd_groups(prob, pat0, motion_pat=None):
    """Column groups for sparse 2-point differences: (groups int32[n], number of groups) -- scipy's ``group_columns`` (same
    column permutation ``RandomState(0).permutation(n)``, same greedy pass) run by ``mvus_group_columns`` on the pattern's
    entries: building the scipy.sparse matrix, converting it to CSC and permuting its columns was 2/3 of a default
    ``Scene.BA`` call's host time at 600 k detections."""
    from . import _lib
    lib = _lib.load()
    n = int(prob.n_params)
    order = np.ascontiguousarray(np.random.RandomState(0).permutation(n), dtype=np.int64)
    groups = np.empty(n, dtype=np.int32)
    # the entries are generated from the codes inside the library (mvus_fd_groups); pattern_entries + mvus_group_columns is the same
    # computation with the entries as arrays (tests/test_fd_mode_host.py compares the three)
    struct, keep = _lib.make_problem_struct(prob)
    pat = np.ascontiguousarray(pat0, dtype=np.int32)
    mp = np.ascontiguousarray(motion_pat, dtype=np.int32) if (prob.motion_reg and motion_pat is not None) else None
    if prob.motion_reg and mp is None:
        raise ValueError('motion_reg: pass the motion-row codes (mvus_ba_motion_pattern)')
    ng = lib.mvus_fd_groups(ctypes.byref(struct), pat.ctypes.data_as(_lib.c_int32_p), mp.ctypes.data_as(_lib.c_int32_p) if mp is not None else None,
                            order.ctypes.data_as(_lib.c_int64_p), groups.ctypes.data_as(_lib.c_int32_p))
    del keep
    if ng < 0:
        raise RuntimeError('mvus_fd_groups failed (%d): %s' % (ng, lib.mvus_last_error(None).decode()))
    return groups, int(ng)


def fd_groups_from_entries(prob, pat0, motion_pat=None):
    """``fd_groups`` with the entries built here (``pattern_entries``) and handed to ``mvus_group_columns``."""
    from . import _lib
    lib = _lib.load()
    rows, cols = pattern_entries(prob, pat0, motion_pat)
    n = int(prob.n_params)
    order = np.ascontiguousarray(np.random.RandomState(0).permutation(n), dtype=np.int64)
    groups = np.empty(n, dtype=np.int32)
    ng = lib.mvus_group_columns(int(prob.n_residuals), n, int(rows.size), rows.ctypes.data_as(_lib.c_int64_p), cols.ctypes.data_as(_lib.c_int64_p),
                                order.ctypes.data_as(_lib.c_int64_p), groups.ctypes.data_as(_lib.c_int32_p))
    if ng < 0:
        raise RuntimeError('mvus_group_columns failed (%d): %s' % (ng, lib.mvus_last_error(None).decode()))
    return groups, int(ng)
