import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvus_amd import spline
rng = np.random.default_rng(0)
for case in range(3):
    m = int(rng.integers(60, 400))
    u = np.cumsum(rng.uniform(0.5, 1.5, m))
    X = np.vstack([10 * np.sin(u / 80 * (1 + case)), 10 * np.cos(u / 95), 30 + 3 * np.sin(u / 50)]) + rng.normal(0, 0.02, (3, m))
s = 1e-4 * (u[-1] - u[0])
print(spline.smooth_fit(u, X, s, full_output=True)[1:])
