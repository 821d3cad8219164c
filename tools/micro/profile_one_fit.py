"""One smoothing fit of the size the incremental loop produces (550k samples, ~1.5k knots) for a rocprofv3 kernel breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mvus_amd import spline, synth
m = int(sys.argv[1]) if len(sys.argv) > 1 else 550000
u = np.linspace(1.0, 1.0 + m / 50.0, m)
rng = np.random.default_rng(0)
X = synth.curve(u) + rng.normal(0, 1e-3, (3, m)) * (rng.uniform(size=m) < 0.02)      # a smooth 50x oversampled curve with a few noisy (triangulated) samples
for s in ([float(a) for a in sys.argv[2:]] or [0.0441, 0.353]):
    spline.smooth_fit(u, X, s)
    t0 = time.perf_counter(); tck = spline.smooth_fit(u, X, s); print('s=%g: %d knots, %.1f ms' % (s, len(tck[0]), 1e3 * (time.perf_counter() - t0)))
