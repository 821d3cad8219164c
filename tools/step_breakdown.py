#!/usr/bin/env python3
"""Wall-clock anatomy of one LM step: Python call vs C++ solve vs sum of GPU kernel time (needs rocprof for the last)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvus_amd import ba, problem as mp, synth
sc = synth.baseline_scene(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
prob, x0 = mp.problem_from_scene(sc)
with ba.BAHandle(prob) as h:
    x = x0.copy()
    for _ in range(3):
        x = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False).x
    torch.cuda.synchronize()
    t0 = time.perf_counter(); inner = 0.0
    n = 20
    for _ in range(n):
        r = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False)
        x = r.x; inner += r.solve_ms
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / n
    print('python wall %.3f ms/step, C++ solve %.3f ms/step' % (wall, inner / n))
