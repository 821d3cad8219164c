"""Where the time of one default Scene.BA (reference algorithm: TRF + LSMR + grouped 2-point differences) goes at configs[1] size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mvus_amd import synth, problem as mp, ba, pattern

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sc = synth.baseline_scene(idx)
prob, x0 = mp.problem_from_scene(sc)
t = time.perf_counter()
h = ba.BAHandle(prob)
print('handle: %.3f s (M=%d n=%d)' % (time.perf_counter() - t, prob.M, x0.size))
for rep in range(2):
    t0 = time.perf_counter(); pat = h.set_pattern(x0); mpat = h.motion_pattern() if h.T else None; t1 = time.perf_counter()
    pn, mn = pattern.resolve_ties(prob, x0, pat, mpat, how='numpy'); t2 = time.perf_counter()
    h.upload_pattern(pn, mn if h.T else None); t3 = time.perf_counter()
    groups, ng = pattern.fd_groups(prob, pn, mn if h.T else None); t4 = time.perf_counter()
    h.set_fd_groups(groups, ng); t5 = time.perf_counter()
    print('rep %d: pattern on GPU + download %.3f s, tie rows in Python %.3f s, upload %.3f s, scipy column groups (%d) %.3f s, upload %.3f s'
          % (rep, t1 - t0, t2 - t1, t3 - t2, ng, t4 - t3, t5 - t4))
    for solver, jm, name in ((ba.SOLVER_TRF_LSMR, ba.JAC_FD, 'trf+fd'), (ba.SOLVER_TRF_LSMR, ba.JAC_PATTERN, 'trf+pattern'), (ba.SOLVER_LM_SCHUR, ba.JAC_ANALYTIC, 'lm')):
        o = ba._lib.default_opts(solver, jm, 10)
        res = ba._lib.MvusResult()
        x = x0.copy()
        import ctypes
        t6 = time.perf_counter()
        h._check(h.lib.mvus_ba_solve(h.h, ba._lib.dptr(x), ctypes.byref(o), ctypes.byref(res), None), 'solve')
        print('   %-12s solve (10 evaluations): %.3f s, %d linear iterations, cost %.6g -> %.6g' % (name, time.perf_counter() - t6, res.lin_iters, res.initial_cost, res.cost))
