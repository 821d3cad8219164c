#!/usr/bin/env python3
"""What the REFERENCE's own BA does to the trajectory at configs[1]'s motion_weights = 1e4 (README of the reference, dataset4), on
this repo's synthetic flight -- from the reference-generated golden fixture tests/golden/config1_shape_7cam.npz (real reference run
by tests/golden/make_golden.py: BA(10) -> remove_outliers -> BA(10 / 200), common.py:441-697, main.py:49-62), no GPU, no reference
import.  For every stored solution: trajectory against the generator's ground truth after the best similarity, the curve's mean
|second difference| (what regulariser F penalises: common.py:959-1001) relative to the true curve's, and the split of the cost into
reprojection and motion terms.  Beside it the same problem solved by the oracle (scipy least_squares, the reference's algorithm)
at motion_weights = 1e2 and 1 -- the weight tools/incremental_loop.py runs the loop with, and why.

    python tools/ref_curve_at_1e4.py            # ~2 min of CPU
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np


def main():
    import golden_cases
    from golden_util import load_case
    from oracle import ba_oracle as orc
    from mvus_amd import bspline
    from mvus_amd.analysis.compare_gt import similarity_from_points
    name = 'config1_shape_7cam'
    sc = golden_cases.make(name)
    scene, g = load_case(name)
    tr = sc.truth
    oprob, ox0 = orc.problem_from_scene(scene)
    assert np.allclose(ox0, g['x0'], rtol=0, atol=1e-12)
    M2 = 2 * sum(d.shape[1] for d in oprob.detections)

    def true_curve(tau):
        X = np.zeros((3, tau.size)); inside = np.zeros(tau.size, dtype=bool)
        for tck in tr['tck']:
            m = (tau >= tck[0][0]) & (tau < tck[0][-1])
            if m.any():
                X[:, m] = bspline.evaluate(tck[0], np.array(tck[1]), tau[m])
            inside |= m
        return X, inside

    def describe(label, prob, x):
        alpha, beta, rs, cams, tck = orc.unpack_x(prob, x)
        f = orc.residual(prob, x)
        ts = np.arange(np.ceil(prob.interval[0, 0]), np.floor(prob.interval[1, -1]), 1.0)
        X = np.zeros((3, ts.size)); ok = np.zeros(ts.size, dtype=bool)
        for s, t in enumerate(tck):
            m = (ts >= prob.interval[0, s]) & (ts <= prob.interval[1, s])
            if m.any():
                X[:, m] = np.array(orc.splev3(ts[m], t)); ok |= m
        Xt, inside = true_curve(ts)
        ok &= inside
        Msim = similarity_from_points(X[:, ok], Xt[:, ok])
        sR, tt = Msim[:3, :3], Msim[:3, 3]
        d = np.sqrt(((Xt[:, ok] - (sR @ X[:, ok] + tt[:, None])) ** 2).sum(axis=0))
        acc = np.linalg.norm(np.diff(sR @ X[:, ok], n=2, axis=1), axis=0).mean()
        acc_t = np.linalg.norm(np.diff(Xt[:, ok], n=2, axis=1), axis=0).mean()
        rep, mot = 0.5 * np.sum(f[:M2] ** 2), 0.5 * np.sum(f[M2:] ** 2)
        nz = f[:M2][f[:M2] != 0]
        print('%-58s rms %7.3f m  max %7.3f m | mean |d2X| %.4f (truth %.4f, ratio %.2f) | cost: reprojection %.4g  motion %.4g | mean |r| %.2f px'
              % (label, np.sqrt(np.mean(d ** 2)), d.max(), acc, acc_t, acc / acc_t, rep, mot, np.mean(np.abs(nz))))

    print('# %s: %d cameras, %d detections, %d parameters, motion_weights %g (fixture: the real reference)' % (name, oprob.C, M2 // 2, ox0.size, oprob.motion_weights))
    describe('start x0 (truth + perturbation)', oprob, g['x0'])
    describe('REFERENCE BA(10), weights 1e4', oprob, g['ba10_x'])
    # the second BA runs on the filtered detections: its x has the same layout (cameras and splines unchanged)
    keep = g['outlier_keep'].astype(bool)
    import copy
    sc2 = copy.deepcopy(scene)
    off = np.concatenate(([0], np.cumsum([d.shape[1] for d in scene.detections])))
    sc2.detections = [d[:, keep[off[i]:off[i + 1]]] for i, d in enumerate(scene.detections)]
    oprob2, _ = orc.problem_from_scene(sc2)
    for tag in ('ba2_10_x', 'ba2_200_x'):
        if tag in g:
            describe('REFERENCE BA(10) -> outliers -> BA(%s), weights 1e4' % tag.split('_')[1], oprob2, g[tag])
    from threadpoolctl import threadpool_limits
    for w in (1e2, 1.0):
        p, x0 = orc.problem_from_scene(scene, motion_weights=w)
        with threadpool_limits(limits=1):
            t0 = time.perf_counter()
            r = orc.solve(p, x0, max_iter=10)
            dt = time.perf_counter() - t0
        describe('oracle (scipy, the reference algorithm) BA(10), weights %g [%.0f s]' % (w, dt), p, r.x)


if __name__ == '__main__':
    main()
