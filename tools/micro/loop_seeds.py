"""The incremental loop (trf) on the sparse 21k flight over several seeds: worst camera-centre error and kept fraction."""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mvus_amd import pipeline, synth
obs = int(sys.argv[1]) if len(sys.argv) > 1 else 21000
for seed in range(2, 10):
    kw = dict(synth.BASELINE_CONFIGS[1]); kw.pop('seed'); kw.pop('num_cam'); kw.pop('total_obs'); kw.pop('num_intervals', None); kw['motion_weights'] = 1e2
    with contextlib.redirect_stdout(io.StringIO()):
        flight, sc = pipeline.staged_scene(7, obs, seed=seed, settings={'ba_solver': 'trf'}, perturb=0.3, **kw)
        pipeline.incremental_reconstruction(flight, max_iter=10)
        ev = pipeline.evaluate_against_truth(flight, sc)
    print('seed %d: max mean err %.3f, traj rms %.3f, max centre %.3f, min kept/clean %.3f' % (seed, max(ev['mean_err']), ev['traj_rms'], max(ev['centre_err']),
          min(k / c for k, c in zip(ev['kept'], ev['clean']))), flush=True)
