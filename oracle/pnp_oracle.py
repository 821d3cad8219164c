"""ORACLE (test infrastructure only -- never imported by the product path).

CPU checker for ``mvus_pnp_ransac`` (the ``cv2.solvePnPRansac`` call of the reference's ``Scene.get_camera_pose``,
``reconstruction/common.py:719-750``).  OpenCV (unpinned, absent from this image) is the reference's implementation of that
step and none of it lives under /root/reference, so there are no reference outputs, golden vectors or tests to pin this
oracle against: **parity unpinned** for the call as a whole.  The one piece of PnP arithmetic the reference itself carries -- its own
six-point direct linear transform ``epipolar.solve_PnP`` (epipolar.py:298-308) -- is run by tests/golden/make_golden_pnp.py and
pins the minimal solver (tests/golden/pnp_dlt.npz).  What this file restates is the published model the call is defined by:

* ``project``       cv2.projectPoints: pinhole projection + the 5-coefficient distortion model (k1 k2 p1 p2 k3);
* ``dlt_pose``      a direct-linear-transform pose from >= 6 correspondences (SVD), the closed form the GPU hypotheses use;
* ``refine``        the pose minimising the pixel reprojection error over a set of points (scipy Levenberg-Marquardt from a
                    start pose) -- what the SOLVEPNP_ITERATIVE refinement inside solvePnPRansac converges to.
"""
import numpy as np


def rodrigues(r):
    r = np.asarray(r, dtype=np.float64).reshape(3)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3) + np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def rotation_to_rvec(R):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(R).as_rotvec()


def distort(x, y, d):
    k1, k2, p1, p2, k3 = d
    r2 = x * x + y * y
    rad = 1 + r2 * (k1 + r2 * (k2 + r2 * k3))
    return x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y


def project(K, d, R, t, X):
    """Pixels (2, N) of X (3, N); depth (N,) returned too."""
    Xc = R @ X + np.asarray(t, dtype=np.float64).reshape(3, 1)
    xd, yd = distort(Xc[0] / Xc[2], Xc[1] / Xc[2], d)
    return np.vstack((K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2])), Xc[2]


def undistort(uv, K, d, iterations=5):
    x0 = (uv[0] - K[0, 2]) / K[0, 0]
    y0 = (uv[1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    k1, k2, p1, p2, k3 = d
    for _ in range(iterations):
        r2 = x * x + y * y
        icd = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * icd, (y0 - dy) * icd
    return np.vstack((x, y))


def dlt_pose(X, xn):
    """Pose (R, t) from 3-D points X (3, n >= 6) and normalised undistorted image coordinates xn (2, n)."""
    n = X.shape[1]
    A = np.zeros((2 * n, 12))
    for p in range(n):
        Xh = np.append(X[:, p], 1.0)
        A[2 * p, 0:4] = Xh
        A[2 * p, 8:12] = -xn[0, p] * Xh
        A[2 * p + 1, 4:8] = Xh
        A[2 * p + 1, 8:12] = -xn[1, p] * Xh
    P = np.linalg.svd(A)[2][-1].reshape(3, 4)
    if np.linalg.det(P[:, :3]) < 0:
        P = -P
    U, s, Vt = np.linalg.svd(P[:, :3])
    R = U @ Vt
    lam = np.trace(R.T @ P[:, :3]) / 3.0
    return R, P[:, 3] / lam


def refine(K, d, R0, t0, X, uv):
    """Minimiser of the pixel reprojection error over all the given points, started at (R0, t0)."""
    from scipy.optimize import least_squares

    def fun(p):
        proj, _ = project(K, d, rodrigues(p[:3]), p[3:], X)
        return np.ravel(proj - uv)

    p0 = np.concatenate((rotation_to_rvec(R0), np.ravel(t0)))
    sol = least_squares(fun, p0, method='lm', xtol=1e-15, ftol=1e-15, gtol=1e-15)
    return rodrigues(sol.x[:3]), sol.x[3:], 0.5 * float(np.sum(sol.fun ** 2))
