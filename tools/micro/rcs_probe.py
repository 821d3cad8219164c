"""One LM step at a BASELINE config with a probe build of the library (MVUS_LIB_PATH): prints whatever the probe kernels printf."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvus_amd import ba, problem as mp, synth
sc = synth.baseline_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
prob, x0 = mp.problem_from_scene(sc)
with ba.BAHandle(prob) as h:
    x = x0.copy()
    for _ in range(2):
        x = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False).x
