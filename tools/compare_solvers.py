#!/usr/bin/env python3
"""Both optimisers of mvus_ba_solve from the same start on a BASELINE config: cost reached and time taken.
   python tools/compare_solvers.py <config> <max_nfev>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvus_amd import ba, problem as mp, synth
cfg, nfev = int(sys.argv[1]), int(sys.argv[2])
prob, x0 = mp.problem_from_scene(synth.baseline_scene(cfg))
for name, solver, jm in (('LM + Schur (analytic J)', ba.SOLVER_LM_SCHUR, ba.JAC_ANALYTIC), ('TRF + LSMR (pattern-masked J)', ba.SOLVER_TRF_LSMR, ba.JAC_PATTERN)):
    with ba.BAHandle(prob) as h:
        h.solve(x0, solver=solver, jac_mode=jm, max_nfev=2, return_fun=False)        # warm-up (allocations)
    with ba.BAHandle(prob) as h:
        t0 = time.perf_counter()
        r = h.solve(x0, solver=solver, jac_mode=jm, max_nfev=nfev, return_fun=False)
        dt = time.perf_counter() - t0
    print('config %d  %-30s nfev %3d njev %3d status %d  cost %.8e -> %.8e  (%.1f ms, %d linear iterations)'
          % (cfg, name, r.nfev, r.njev, r.status, r.initial_cost, r.cost, 1e3 * dt, r.lin_iters))
