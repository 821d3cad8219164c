"""Alignment of the reconstructed trajectory with a ground-truth track (reference ``analysis/compare_gt.py``).

Host numpy / scipy: this runs once after the reconstruction, on a few thousand ground-truth samples.  Same steps and
names as the reference (``align_gt`` compare_gt.py:73-151, ``optimize`` :33-70, ``error_M`` :21-30):

  1. resample the spline at the ground-truth rate, slide it over the ground truth frame by frame and keep the offset
     with the smallest mean distance after a similarity fit (coarse search, :104-123);
  2. refine (alpha, beta) of ``t_gt = alpha * k + beta`` with a Cauchy-loss least squares on the per-sample distances
     after the similarity fit (:126, :33-70);
  3. drop samples farther than 10x the mean error and report mean / median distance (:128-137).

The similarity fit restates ``thirdparty/transformation.py:affine_matrix_from_points(shear=False, scale=True)``:
rotation by Kabsch (SVD of the covariance), uniform scale = ratio of the RMS distances from the centroids.
"""
import numpy as np
from scipy.optimize import least_squares

from ..tools import util


def similarity_from_points(v0, v1):
    """4x4 similarity M with M [v0; 1] ~ [v1; 1] (3 x k point sets): Kabsch rotation, RMS-ratio scale."""
    v0 = np.array(v0, dtype=np.float64, copy=True)
    v1 = np.array(v1, dtype=np.float64, copy=True)
    if v0.shape != v1.shape or v0.shape[0] != 3 or v0.shape[1] < 3:
        raise ValueError('input arrays are of wrong shape or type')
    c0, c1 = v0.mean(axis=1, keepdims=True), v1.mean(axis=1, keepdims=True)
    a, b = v0 - c0, v1 - c1
    u, s, vh = np.linalg.svd(b @ a.T)
    R = u @ vh
    if np.linalg.det(R) < 0.0:                       # keep a right-handed system
        R -= np.outer(u[:, 2], vh[2, :] * 2.0)
    R = R * np.sqrt(np.sum(b * b) / np.sum(a * a))
    M = np.identity(4)
    M[:3, :3] = R
    M[:3, 3:] = c1 - R @ c0
    return M


def error_M(model, data, param=None):
    """Distances between the transformed reconstruction data[:3] and the ground truth data[3:]."""
    M = np.asarray(model).reshape(4, 4)
    tran = M @ util.homogeneous(data[:3])
    tran = tran / tran[-1]
    return np.sqrt(np.sum((data[3:] - tran[:3]) ** 2, axis=0))


def optimize(alpha, beta, flight, gt):
    """Fine alignment of the ground-truth clock (compare_gt.py:33-70).  Returns (least_squares result,
    (transformed reconstruction [4,k], ground truth [3,k], M, distances [k]))."""

    def error_fn(model, output=False):
        a, b = model[0], model[1]
        t_gt = a * np.arange(gt.shape[1]) + b if gt.shape[0] == 3 else a * (gt[0] - gt[0, 0]) + b
        _, idx = util.sampling(t_gt, flight.spline['int'])
        gt_part = gt[-3:, idx]
        traj = flight.spline_to_traj(t=t_gt[idx])
        data = np.vstack((traj[1:], gt_part))
        M = similarity_from_points(traj[1:], gt_part)
        dist = error_M(M.ravel(), data)
        if output:
            tran = M @ util.homogeneous(traj[1:])
            tran = tran / tran[-1]
            return np.vstack((traj[0], tran[:3])), gt_part, M, dist
        error = np.zeros(gt.shape[1], dtype=float)
        error[idx] = dist
        return error

    ls = least_squares(error_fn, np.array([alpha, beta], dtype=float), loss='cauchy', f_scale=1)
    return ls, error_fn(ls.x, output=True)


def align_gt(flight, f_gt, gt_path, visualize=False, verbose=True):
    """Align ``flight`` (a Scene with a spline) with the ground truth in ``gt_path`` (text file, 3 rows x,y,z or 4 rows
    t,x,y,z, either orientation; or the array itself) sampled at ``f_gt`` Hz.  Returns the reference's dict
    (align_param, reconst_tran, gt, tran_matrix, error) or None when no ground truth is given."""
    if isinstance(gt_path, np.ndarray):
        gt_ori = np.asarray(gt_path, dtype=np.float64)
    else:
        if not len(gt_path):
            print('No ground truth data provided\n')
            return None
        try:
            gt_ori = np.loadtxt(gt_path)
        except Exception:
            print('Ground truth not correctly loaded')
            return None
    if gt_ori.shape[0] in (3, 4):
        pass
    elif gt_ori.shape[1] in (3, 4):
        gt_ori = gt_ori.T
    else:
        raise Exception('Ground truth data have an invalid shape')
    if visualize:
        raise NotImplementedError('plotting is outside this package (SURVEY.md section 2)')

    alpha = flight.cameras[flight.settings['ref_cam']].fps / f_gt
    reconst = flight.spline_to_traj(sampling_rate=alpha)
    t0 = reconst[0, 0]
    reconst = np.vstack(((reconst[0] - t0) / alpha, reconst[1:]))
    gt = np.vstack((np.arange(gt_ori.shape[1]), gt_ori)) if gt_ori.shape[0] == 3 else np.vstack((gt_ori[0] - gt_ori[0, 0], gt_ori[1:]))

    thres = int(reconst[0, -1] / 2)                          # coarse search over whole-frame offsets
    if int(gt[0, -1] - thres) < 0:
        raise Exception('Ground truth too short!')
    error_min, j = np.inf, 0
    for i in range(-thres, int(gt[0, -1] - thres)):
        p1, p2 = util.match_overlap(np.vstack((reconst[0] + i, reconst[1:])), gt)
        M = similarity_from_points(p1[1:], p2[1:])
        err = np.mean(error_M(M.ravel(), np.vstack((p1[1:], p2[1:]))))
        if err < error_min:
            error_min, j = err, i
    beta = t0 - alpha * j

    ls, res = optimize(alpha, beta, flight, gt_ori)
    error_ = res[3]
    idx = error_ <= 10 * np.mean(error_)                     # relative outlier threshold
    out = {'align_param': ls.x, 'reconst_tran': res[0][:, idx], 'gt': res[1][:, idx], 'tran_matrix': res[2], 'error': error_[idx]}
    if verbose:
        print('The mean error (distance) is {:.5f} meter\n'.format(np.mean(out['error'])))
        print('The median error (distance) is {:.5f} meter\n'.format(np.median(out['error'])))
    return out
