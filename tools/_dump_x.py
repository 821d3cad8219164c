import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from mvus_amd import ba, problem as mp, synth
sc = synth.baseline_scene(2)
prob, x0 = mp.problem_from_scene(sc)
with ba.BAHandle(prob) as h:
    x = x0.copy()
    for i in range(10):
        x = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False).x
    np.save('gpurun_out/x10.npy', x)
    f, J, span = h.residual_jacobian(x)[:3] if False else (None, None, None)
