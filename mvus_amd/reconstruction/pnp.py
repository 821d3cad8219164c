"""Absolute camera pose from 3-D / 2-D correspondences on the GPU: the ``cv2.solvePnPRansac`` call of the reference's
``Scene.get_camera_pose`` (common.py:719-750).  OpenCV is not part of this image, so the reference's output cannot be
reproduced number for number (parity unpinned); ``mvus_pnp_ransac`` restates the call's contract -- RANSAC over minimal-sample
poses scored by reprojection error, then least-squares refinement on the inliers -- and is checked against ground truth and an
independent minimiser (tests/test_pnp.py).  No CPU fallback."""
import ctypes

import numpy as np

from .. import _lib


def solve_pnp_ransac(object_points, image_points, K, d, reprojectionError=8.0, iterationsCount=100, seed=0, device=0):
    """``cv2.solvePnPRansac(objectPoints, imagePoints, K, d, reprojectionError=...)``: object_points (N, 3) or (N, 1, 3),
    image_points (N, 2) or (N, 1, 2) raw pixels, K 3x3, d the 5 distortion coefficients.  Returns (retval, rvec (3, 1),
    tvec (3, 1), inliers (n, 1) int32 indices) like OpenCV; retval False when no pose is supported by six points."""
    lib = _lib.load()
    X = np.asarray(object_points, dtype=np.float64).reshape(-1, 3)
    uv = np.asarray(image_points, dtype=np.float64).reshape(-1, 2)
    if X.shape[0] != uv.shape[0]:
        raise ValueError('object and image points differ in number')
    K = np.asarray(K, dtype=np.float64)
    Kv = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
    dv = np.zeros(5)
    dd = np.ravel(np.asarray(d, dtype=np.float64)) if d is not None else np.zeros(0)
    dv[:min(5, dd.size)] = dd[:5]
    N = X.shape[0]
    Xs = np.ascontiguousarray(X.T)
    uvs = np.ascontiguousarray(uv.T)
    rvec, tvec = np.zeros(3), np.zeros(3)
    mask = np.zeros(max(N, 1), dtype=np.uint8)
    n_in = ctypes.c_int64(0)
    rc = lib.mvus_pnp_ransac(int(device), N, _lib.dptr(Xs), _lib.dptr(uvs), _lib.dptr(Kv), _lib.dptr(dv), float(reprojectionError),
                             int(iterationsCount), int(seed), _lib.dptr(rvec), _lib.dptr(tvec), mask.ctypes.data_as(_lib.c_uint8_p),
                             ctypes.byref(n_in))
    if rc == _lib.MVUS_E_NUMERIC:
        return False, None, None, None
    if rc != 0:
        raise (ValueError if rc == _lib.MVUS_E_INVALID else RuntimeError)('mvus_pnp_ransac: %s' % lib.mvus_last_error(None).decode())
    inliers = np.flatnonzero(mask[:N]).astype(np.int32).reshape(-1, 1)
    return True, rvec.reshape(3, 1), tvec.reshape(3, 1), inliers
