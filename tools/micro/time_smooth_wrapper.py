"""Where the time of spline.smooth_fit goes outside the C call (3.3 M samples)."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mvus_amd import spline, synth
m = 3278873
u = np.linspace(1.0, 1.0 + m / 50.0, m)
rng = np.random.default_rng(0)
X = synth.curve(u) + rng.normal(0, 1e-3, (3, m)) * (rng.uniform(size=m) < 0.02)
spline.smooth_fit(u, X, 8.0)
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    spline.smooth_fit(u, X, 8.0)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(12); print(s.getvalue()[:3000])
