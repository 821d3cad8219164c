"""Hunt for reads of uninitialised device memory: fill a few GB with NaN / huge values through torch, hand the memory back to
HIP, then create a handle and solve (fresh hipMalloc blocks now hold the poison instead of zeros)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mvus_amd import synth, problem as mp, ba
for index in (1, 4, 0):
    sc = synth.baseline_scene(index); prob, x0 = mp.problem_from_scene(sc)
    outcomes = collections.Counter()
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
        val = float('nan') if rep % 2 == 0 else 1e300
        junk = [torch.full((1 << 27,), val, dtype=torch.float64, device='cuda') for _ in range(4)]      # 4 GiB
        torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
        with ba.BAHandle(prob) as h:
            r = h.solve(x0, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=8)
            r2 = h.solve(x0, solver=ba.SOLVER_TRF_LSMR, jac_mode=ba.JAC_PATTERN, max_nfev=3)
            outcomes[(round(r.cost, 2), r.nfev, r.status, r.cost < r.initial_cost, round(r2.cost, 2))] += 1
    print('config', index, dict(outcomes))
