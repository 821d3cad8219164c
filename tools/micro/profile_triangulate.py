"""cProfile of the incremental loop's host side (where do triangulate's seconds go?)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvus_amd import pipeline, synth
kw = dict(synth.BASELINE_CONFIGS[1]); kw.pop('seed'); kw.pop('num_cam'); kw.pop('total_obs'); kw.pop('num_intervals', None); kw['motion_weights'] = 100.0
flight, sc = pipeline.staged_scene(7, int(sys.argv[1]) if len(sys.argv) > 1 else 100000, seed=2, settings={'ba_solver': sys.argv[2] if len(sys.argv) > 2 else 'trf'}, perturb=0.3, **kw)
pr = cProfile.Profile(); pr.enable()
timer = pipeline.incremental_reconstruction(flight, max_iter=10)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
print({k: round(v, 2) for k, v in timer.totals().items()})
