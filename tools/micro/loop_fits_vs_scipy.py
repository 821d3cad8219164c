"""Every smoothing fit of the incremental loop (trf, sparse flight) against scipy's splprep on the same input."""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy import interpolate
from mvus_amd import pipeline, synth, spline
obs = int(sys.argv[1]) if len(sys.argv) > 1 else 21000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
Base = spline.SmoothFit
class Checked(Base):                                       # traj_fit fits through a SmoothFit session: check every call of it
    def __init__(self, t, X, device=0):
        super().__init__(t, X, device=device)
        self._t, self._X = np.asarray(t, dtype=np.float64).copy(), np.asarray(X, dtype=np.float64).copy()
    def __call__(self, s, full_output=False):
        out = super().__call__(s, full_output=True)
        tck, fp, ier = out
        t, X = self._t, self._X
        (tck0, _), fp0, ier0, _ = interpolate.splprep(X, u=t, s=s, k=3, full_output=True)
        t0, c0 = tck0[0], tck0[1]
        same = len(t0) == len(tck[0]) and np.array_equal(t0, tck[0])
        dc = float(np.max(np.abs(np.asarray(c0) - np.asarray(tck[1])))) if same else float('nan')
        rows.append((t.size, s, len(tck[0]), len(t0), same, dc, fp, fp0, ier, ier0))
        if not same and not os.path.exists('gpurun_out/mismatch_fit.npz'):
            os.makedirs('gpurun_out', exist_ok=True)
            np.savez_compressed('gpurun_out/mismatch_fit.npz', t=t, X=X, s=s, knots_gpu=tck[0], knots_scipy=t0)
        return out if full_output else tck
spline.SmoothFit = Checked
kw = dict(synth.BASELINE_CONFIGS[1]); kw.pop('seed'); kw.pop('num_cam'); kw.pop('total_obs'); kw.pop('num_intervals', None); kw['motion_weights'] = 1e2
with contextlib.redirect_stdout(io.StringIO()):
    flight, sc = pipeline.staged_scene(7, obs, seed=seed, settings={'ba_solver': 'trf'}, perturb=0.3, **kw)
    pipeline.incremental_reconstruction(flight, max_iter=10)
    ev = pipeline.evaluate_against_truth(flight, sc)
for r in rows:
    print('m=%d s=%.4g knots %d / scipy %d same=%s dcoef %.2e fp %.6g / %.6g ier %d / %d' % r)
print('max centre %.3f' % max(ev['centre_err']))
