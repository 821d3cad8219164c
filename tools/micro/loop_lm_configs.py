import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvus_amd import synth, problem as mp, ba
for index in (1, 4):
    sc = synth.baseline_scene(index); prob, x0 = mp.problem_from_scene(sc)
    outcomes = collections.Counter()
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
        with ba.BAHandle(prob) as h:
            r = h.solve(x0, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=8)
            outcomes[(round(r.cost, 3), r.nfev, r.njev, r.status, r.cost < r.initial_cost)] += 1
    print('config', index, dict(outcomes))
