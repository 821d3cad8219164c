import os, sys, time
sys.path.insert(0, '' + os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + '')
import torch, numpy as np
from mvus_amd import ba, problem as mp, synth, _lib
sc = synth.baseline_scene(2)
prob, x0 = mp.problem_from_scene(sc)
with ba.BAHandle(prob) as h:
    x = x0.copy()
    for _ in range(3):
        x = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False).x
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter(); inner = 0.0
    for _ in range(n):
        r = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False)
        x = r.x; inner += r.solve_ms
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / n
    print('python wall %.3f ms/step, C++ solve %.3f ms/step' % (wall, inner / n))
    # one long call
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=31, return_fun=False)
    torch.cuda.synchronize()
    print('long solve: %.3f ms per trial (%d trials, %d linearisations)' % ((time.perf_counter() - t0) * 1e3 / (r.nfev - 1), r.nfev - 1, r.njev))
