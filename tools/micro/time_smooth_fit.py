"""Wall time of mvus_spline_smooth at the sample counts the incremental loop feeds it (50x oversampled trajectories)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mvus_amd import spline
for m in (6000, 60000, 560000):
    u = np.arange(m) * (11000.0 / m)
    X = np.vstack((10 * np.sin(u / 80), 10 * np.cos(u / 95), 30 + 3 * np.sin(u / 50))) + np.random.default_rng(0).normal(0, 0.002, (3, m))
    for s in (1e-6 * 11000, 1e-3 * 11000 * (m / 11000.0) * 4e-3, ):
        spline.smooth_fit(u, X, s)
        t0 = time.perf_counter(); tck = spline.smooth_fit(u, X, s); dt = time.perf_counter() - t0
        print('m %7d s %.3g: %4d knots, %.1f ms' % (m, s, len(tck[0]), dt * 1e3))
