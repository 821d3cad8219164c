"""LAPACK reference for one damped Gauss-Newton step of the LM + Schur solver, from the normal-equation blocks the
library exports (mvus_ba_normal_equations): banded Cholesky of the spline block, dense Schur complement on the camera
block.  Test infrastructure (checks mvus_ba_lm_step, i.e. the whole GPU solve chain)."""
import numpy as np


def lapack_lm_step(prob, g, A, band, cross, lam):
    """p = -(H + lam * D)^-1 g with H = [[A, E], [E^T, C]] given as camera blocks A[C,B,B], block band[N,W,3,3] and
    cross[C,B,3N]; D = diag(H) (1 where 0)."""
    from scipy.linalg import solveh_banded
    C, B, N, W = prob.C, 3 + prob.P, band.shape[0], band.shape[1]
    cam_cols = np.array([[c, C + c, 2 * C + c] + list(range(3 * C + c * prob.P, 3 * C + (c + 1) * prob.P)) for c in range(C)])
    spl_cols = np.concatenate([[int(prob.spline_x_offsets[s_]) + d * int(n_) + j for j in range(int(n_)) for d in range(3)]
                               for s_, n_ in enumerate(prob.n_coef)])
    bw = 3 * W - 1
    ab = np.zeros((bw + 1, 3 * N))                                  # upper banded storage of the spline block
    for w in range(W):
        for a_ in range(3):
            for b_ in range(3):
                off = 3 * w + b_ - a_
                if off < 0:
                    continue
                rows = 3 * np.arange(N - w) + a_
                ab[bw - off, rows + off] = band[:N - w, w, a_, b_]
    dS = ab[bw].copy()
    ab[bw] += lam * np.where(dS > 0, dS, 1.0)
    Esp = cross.reshape(C * B, 3 * N)
    Z = solveh_banded(ab, np.column_stack([Esp.T, g[spl_cols]]))
    Acam = np.zeros((C * B, C * B))
    for c in range(C):
        Acam[c * B:(c + 1) * B, c * B:(c + 1) * B] = A[c]
    dA = np.diag(Acam).copy()
    Sred = Acam + lam * np.diag(np.where(dA > 0, dA, 1.0)) - Esp @ Z[:, :-1]
    pc = -np.linalg.solve(Sred, g[cam_cols.ravel()] - Esp @ Z[:, -1])
    ps = -(Z[:, -1] + Z[:, :-1] @ pc)
    p = np.zeros(prob.n_params)
    p[cam_cols.ravel()] = pc
    p[spl_cols] = ps
    return p
