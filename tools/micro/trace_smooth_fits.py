"""Every smoothing fit of the incremental loop: samples, smoothing factor, knots found, wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvus_amd import pipeline, synth, spline
kw = dict(synth.BASELINE_CONFIGS[1]); kw.pop('seed'); kw.pop('num_cam'); kw.pop('total_obs'); kw.pop('num_intervals', None); kw['motion_weights'] = 100.0
flight, sc = pipeline.staged_scene(7, int(sys.argv[1]) if len(sys.argv) > 1 else 100000, seed=2, settings={'ba_solver': 'lm'}, perturb=0.3, **kw)
Base = spline.SmoothFit
class Traced(Base):                                        # traj_fit fits through a SmoothFit session
    def __init__(self, t, X, device=0):
        t0 = time.perf_counter(); super().__init__(t, X, device=device)
        self._span = (float(t[0]), float(t[-1]))
        print('open m=%d: %.1f ms' % (self.m, (time.perf_counter() - t0) * 1e3), flush=True)
    def __call__(self, s, full_output=False):
        t0 = time.perf_counter(); out = super().__call__(s, full_output=full_output); dt = time.perf_counter() - t0
        tck = out[0] if full_output else out
        print('fit m=%d span=%.0f s=%.3g -> %d knots, %.1f ms' % (self.m, self._span[1] - self._span[0], s, len(tck[0]), dt * 1e3), flush=True)
        return out
spline.SmoothFit = Traced
pipeline.incremental_reconstruction(flight, max_iter=10)
