"""The one function of the reference's ``reconstruction/epipolar.py`` that sits next to the BA hot path
(SURVEY.md 8f rank 4): ``triangulate_matlab`` (epipolar.py:497-510), called per camera pair by ``Scene.triangulate``
(common.py:783) -- there a Python loop with one 4x4 ``np.linalg.svd`` per point, here one lane per point in
``k_triangulate`` (csrc/triangulate.hip.h, one-sided Jacobi SVD) through ``mvus_triangulate`` of the C ABI.
Fundamental-matrix estimation, RANSAC and PnP stay out of scope.  No CPU fallback."""
import ctypes

import numpy as np

from .. import _lib


def _call(x1, x2, P1, P2, errors, device):
    lib = _lib.load()
    x1 = np.ascontiguousarray(np.asarray(x1, dtype=np.float64)[:2])
    x2 = np.ascontiguousarray(np.asarray(x2, dtype=np.float64)[:2])
    if x1.shape != x2.shape or x1.ndim != 2:
        raise ValueError('x1 and x2 must both be (2 or 3) x N')
    P1 = np.ascontiguousarray(P1, dtype=np.float64)
    P2 = np.ascontiguousarray(P2, dtype=np.float64)
    if P1.shape != (3, 4) or P2.shape != (3, 4):
        raise ValueError('P1 and P2 must be 3 x 4')
    N = x1.shape[1]
    X = np.empty((4, N))
    e1 = np.empty(N) if errors else None
    e2 = np.empty(N) if errors else None
    rc = lib.mvus_triangulate(int(device), N, _lib.dptr(x1), _lib.dptr(x2), _lib.dptr(P1), _lib.dptr(P2), _lib.dptr(X),
                              _lib.dptr(e1) if errors else None, _lib.dptr(e2) if errors else None)
    if rc != 0:
        raise (ValueError if rc == _lib.MVUS_E_INVALID else RuntimeError)('mvus_triangulate: ' + lib.mvus_last_error(None).decode())
    return X, e1, e2


def triangulate_matlab(x1, x2, P1, P2, device=0):
    """x1, x2: (2 or 3) x N pixel coordinates (a homogeneous third row is ignored, like the reference only reads rows 0
    and 1); returns 4 x N homogeneous points with last row 1 (epipolar.py:497-510)."""
    return _call(x1, x2, P1, P2, False, device)[0]


def triangulate_with_errors(x1, x2, P1, P2, device=0):
    """triangulate_matlab plus the reprojection distances in both cameras (what Scene.triangulate thresholds,
    common.py:786-789), computed in the same kernel."""
    return _call(x1, x2, P1, P2, True, device)


def reprojection_error(x, x_p):
    """epipolar.py:639."""
    return np.sqrt((x[0] - x_p[0]) ** 2 + (x[1] - x_p[1]) ** 2)
