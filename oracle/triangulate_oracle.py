"""CPU ORACLE for the triangulation step  --  TEST INFRASTRUCTURE ONLY (see oracle/ba_oracle.py for the rules).

numpy restatement of ``epipolar.triangulate_matlab`` (reference reconstruction/epipolar.py:497-510),
``epipolar.reprojection_error`` (:639) and ``Camera.projectPoint`` (common.py:1072-1079).  Pinned against the
reference itself: tests/golden/make_golden_triangulate.py imports the real functions and stores their outputs
(tests/golden/triangulate_2cam.npz); tests/test_triangulate.py checks this file against them.
Third-party arithmetic: ``numpy.linalg.svd`` (LAPACK gesdd), called exactly like the reference calls it."""
import numpy as np


def triangulate_matlab(x1, x2, P1, P2):
    X = np.zeros((4, x1.shape[1]))
    for i in range(x1.shape[1]):
        r1 = x1[0, i] * P1[2] - P1[0]
        r2 = x1[1, i] * P1[2] - P1[1]
        r3 = x2[0, i] * P2[2] - P2[0]
        r4 = x2[1, i] * P2[2] - P2[1]
        A = np.array([r1, r2, r3, r4])
        U, S, V = np.linalg.svd(A)
        X[:, i] = V[-1] / V[-1, -1]
    return X


def project(P, X):
    """Camera.projectPoint: X 3xN or 4xN -> 3xN with last row 1."""
    if X.shape[0] == 3:
        X = np.vstack((X, np.ones(X.shape[1])))
    x = P @ X
    return x / x[2]


def reprojection_error(x, x_p):
    return np.sqrt((x[0] - x_p[0]) ** 2 + (x[1] - x_p[1]) ** 2)
