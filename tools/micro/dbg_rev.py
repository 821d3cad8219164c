import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from mvus_amd import ba, problem as mp, synth
from test_gpu_det_assembly import _reverse_camera
for cfg in (0, 1):
    kw = dict(synth.BASELINE_CONFIGS[cfg])
    if cfg == 1: kw.update(total_obs=20_000)
    prob, x0 = mp.problem_from_scene(synth.make_scene(**kw))
    rprob, rows = _reverse_camera(prob, 1)
    for solver, jac, name in ((ba.SOLVER_LM_SCHUR, ba.JAC_ANALYTIC, 'lm'), (ba.SOLVER_TRF_LSMR, ba.JAC_PATTERN, 'trf')):
        for nfev in (2, 3, 4, 6):
            out = []
            for p in (prob, rprob, prob):
                with ba.BAHandle(p) as h:
                    r = h.solve(x0, solver=solver, jac_mode=jac, max_nfev=nfev)
                    out.append(r.cost)
            os.environ['MVUS_ASM_ATOMIC'] = '1'
            with ba.BAHandle(prob) as h:
                out.append(h.solve(x0, solver=solver, jac_mode=jac, max_nfev=nfev).cost)
            os.environ.pop('MVUS_ASM_ATOMIC')
            print(cfg, name, nfev, ['%.12e' % c for c in out], flush=True)
