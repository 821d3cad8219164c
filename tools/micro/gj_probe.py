import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvus_amd import synth, problem as mp, ba
sc = synth.baseline_scene(2); prob, x0 = mp.problem_from_scene(sc)
h = ba.BAHandle(prob)
r = h.solve(x0, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=3, return_fun=False)
