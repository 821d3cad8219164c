import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy import interpolate
from mvus_amd import spline
rng = np.random.default_rng(0)
bad = 0
for case in range(8):
    m = int(rng.integers(60, 400)) if case < 7 else 5000
    u = np.cumsum(rng.uniform(0.5, 1.5, m))
    X = np.vstack([10 * np.sin(u / 80 * (1 + case % 7)), 10 * np.cos(u / 95), 30 + 3 * np.sin(u / 50)]) + rng.normal(0, 0.02, (3, m))
    for s in [1e-6 * (u[-1] - u[0]), 1e-4 * (u[-1] - u[0]), 0.05, 1.0, 50.0]:
        t0 = time.time(); ((tk, c0, _), _), fp0, ier0, _ = interpolate.splprep(X, u=u, s=s, k=3, full_output=1); t_sp = time.time() - t0
        t0 = time.time(); tck, fp, ier = spline.smooth_fit(u, X, s, full_output=True); t_gpu = time.time() - t0
        same = tck[0].size == tk.size and np.array_equal(tck[0], tk)
        dc = max(np.max(np.abs(tck[1][d] - c0[d])) for d in range(3)) if same else float('nan')
        bad += (not same) or not (dc < 1e-6)
        print(case, m, '%.3g' % s, 'n', tk.size, tck[0].size, 'same', same, 'dc %.2e' % dc, 'fp %.6g %.6g' % (fp0, fp), 'ier', ier0, ier, 'scipy %.3fs gpu %.3fs' % (t_sp, t_gpu))
print('BAD', bad)
