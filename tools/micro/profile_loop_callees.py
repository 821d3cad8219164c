import cProfile, pstats, os, sys, io
sys.path.insert(0, '/root/repo')
from mvus_amd import pipeline, synth
kw = dict(synth.BASELINE_CONFIGS[1]); kw.pop('seed'); kw.pop('num_cam'); kw.pop('total_obs'); kw.pop('num_intervals', None); kw['motion_weights'] = 100.0
flight, sc = pipeline.staged_scene(7, 100000, seed=2, settings={'ba_solver': 'trf'}, perturb=0.3, **kw)
pr = cProfile.Profile(); pr.enable()
timer = pipeline.incremental_reconstruction(flight, max_iter=10)
pr.disable()
s = io.StringIO(); st = pstats.Stats(pr, stream=s); st.sort_stats('cumulative').print_callees('common.py:.*\\(BA\\)'); print(s.getvalue()[:5000])
s = io.StringIO(); st = pstats.Stats(pr, stream=s); st.sort_stats('cumulative').print_callees('common.py:.*\\(triangulate\\)'); print(s.getvalue()[:3500])
