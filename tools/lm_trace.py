import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mvus_amd import ba, _lib, problem as mp, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sc = synth.baseline_scene(cfg)
prob, x0 = mp.problem_from_scene(sc)
opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, int(sys.argv[2]) if len(sys.argv) > 2 else 12)
opts.verbose = 2
with ba.BAHandle(prob) as h:
    r = h.solve(x0, opts=opts)
    print('status', r.status, 'nfev', r.nfev, 'cost', r.initial_cost, '->', r.cost)
    f = h.residual(x0)
    print('f parts: det', 0.5 * np.sum(f[:2 * prob.M] ** 2), 'motion', 0.5 * np.sum(f[2 * prob.M:] ** 2))
