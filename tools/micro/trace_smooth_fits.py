"""Every smoothing fit of the incremental loop: samples, smoothing factor, knots found, wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvus_amd import pipeline, synth, spline
kw = dict(synth.BASELINE_CONFIGS[1]); kw.pop('seed'); kw.pop('num_cam'); kw.pop('total_obs'); kw.pop('num_intervals', None); kw['motion_weights'] = 100.0
flight, sc = pipeline.staged_scene(7, int(sys.argv[1]) if len(sys.argv) > 1 else 100000, seed=2, settings={'ba_solver': 'lm'}, perturb=0.3, **kw)
orig = spline.smooth_fit
def traced(t, X, s, device=0, full_output=False):
    t0 = time.perf_counter(); out = orig(t, X, s, device=device, full_output=full_output); dt = time.perf_counter() - t0
    tck = out[0] if full_output else out
    print('fit m=%d span=%.0f s=%.3g -> %d knots, %.1f ms' % (t.size, t[-1] - t[0], s, len(tck[0]), dt * 1e3), flush=True)
    return out
spline.smooth_fit = traced
pipeline.incremental_reconstruction(flight, max_iter=10)
