#!/usr/bin/env python3
"""Timings of the steps either side of BA on the GPU against the host libraries the reference calls:
scipy's splprep (FITPACK) for traj_to_spline; the PnP step has no host counterpart in this image (OpenCV absent).
    python tools/time_neighbours.py > profiles/r02_neighbour_steps.txt        (on the GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy import interpolate
from mvus_amd import spline
from mvus_amd.reconstruction.pnp import solve_pnp_ransac


def trajectory(seed, m, noise=0.02):
    rng = np.random.default_rng(seed)
    u = np.cumsum(rng.uniform(0.5, 1.5, m))
    X = np.vstack([10 * np.sin(u / 80), 10 * np.cos(u / 95), 30 + 3 * np.sin(u / 50)]) + rng.normal(0, noise, (3, m))
    return u, X


def best(fn, reps=3):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); t.append(time.perf_counter() - t0)
    return min(t), out


print('# smoothing-spline fit: scipy.interpolate.splprep (FITPACK, host) vs mvus_spline_smooth (GPU), same knots required')
print('%8s %12s %8s %12s %12s %8s' % ('samples', 's', 'knots', 'scipy_s', 'gpu_s', 'speedup'))
spline.smooth_fit(*trajectory(0, 100), 1.0)
for m in (500, 2000, 5000, 18000):
    u, X = trajectory(m, m)
    for s in (1e-6 * (u[-1] - u[0]), 0.05 * m / 500, 3 * m * 0.02 ** 2):
        tg, tck = best(lambda: spline.smooth_fit(u, X, s))
        if m <= 5000:
            ts, ref = best(lambda: interpolate.splprep(X, u=u, s=s, k=3)[0], reps=1)
            assert np.array_equal(ref[0], tck[0])
            print('%8d %12.4g %8d %12.4f %12.4f %8.1f' % (m, s, tck[0].size, ts, tg, ts / tg))
        else:
            print('%8d %12.4g %8d %12s %12.4f %8s' % (m, s, tck[0].size, '(minutes)', tg, '-'))
part = np.vstack(trajectory(3, 6000))
t0 = time.perf_counter(); tck = spline.traj_fit(part, [10, 20]); t1 = time.perf_counter() - t0
print('# traj_fit (the whole smooth_factor loop of traj_to_spline) on 6000 samples: %.3f s, %d knots' % (t1, tck[0].size))

print('# PnP + RANSAC (mvus_pnp_ransac), 100 hypotheses, 10 % gross outliers')
K = np.array([[1100.0, 0, 960], [0, 1080.0, 540], [0, 0, 1]])
for N in (600, 6000, 60000):
    rng = np.random.default_rng(N)
    t = np.linspace(0, 600, N)
    X = np.vstack((10 * np.sin(t / 80), 10 * np.cos(t / 95), 30 + 3 * np.sin(t / 50)))
    R = np.eye(3); tv = np.array([1.0, -2.0, 15.0])
    Xc = R @ X + tv.reshape(3, 1)
    uv = np.vstack((K[0, 0] * Xc[0] / Xc[2] + K[0, 2], K[1, 1] * Xc[1] / Xc[2] + K[1, 2])) + rng.normal(0, 0.5, (2, N))
    bad = rng.random(N) < 0.1
    uv[:, bad] += 100.0
    tg, out = best(lambda: solve_pnp_ransac(X.T, uv.T, K, np.zeros(5)))
    print('%8d points: %.4f s, %d inliers, |t - truth| = %.2e' % (N, tg, out[3].shape[0], np.linalg.norm(np.ravel(out[2]) - tv)))
