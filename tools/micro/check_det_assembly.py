"""Deterministic assembly (MVUS_DET_ASSEMBLY=1): the normal equations against the atomic assembly, bit-reproducibility of the
assembly and of a whole LM solve from run to run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mvus_amd import ba, problem as mp, synth

def run(cfg, det, reps=3):
    if det: os.environ['MVUS_DET_ASSEMBLY'] = '1'
    else: os.environ.pop('MVUS_DET_ASSEMBLY', None)
    kw = dict(synth.BASELINE_CONFIGS[cfg])
    scene = synth.make_scene(**kw)
    prob, x0 = mp.problem_from_scene(scene)
    out = []
    for r in range(reps):
        with ba.BAHandle(prob) as h:
            h.residual_jacobian(x0)
            g, cam, band, cross = h.normal_equations()
            res = h.solve(x0, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=8)
            out.append((g, cam, band, cross, res.cost, res.x.copy()))
    return out

for cfg in [int(a) for a in sys.argv[1:]] or [2]:
    a = run(cfg, False); d = run(cfg, True)
    for name, i in (('g', 0), ('cam', 1), ('band', 2), ('cross', 3)):
        ref = a[0][i]; scale = np.max(np.abs(ref)) + 1e-300
        print('config %d %-5s: det vs atomic max |diff| / max |ref| = %.2e; atomic run-to-run identical: %s; det run-to-run identical: %s'
              % (cfg, name, np.max(np.abs(d[0][i] - ref)) / scale, all(np.array_equal(a[0][i], a[k][i]) for k in range(1, len(a))),
                 all(np.array_equal(d[0][i], d[k][i]) for k in range(1, len(d)))))
    print('config %d LM cost after 8 evaluations: atomic %s ; det %s' % (cfg, [repr(o[4]) for o in a], [repr(o[4]) for o in d]))
    print('config %d LM x identical run to run: atomic %s, det %s' % (cfg, all(np.array_equal(a[0][5], o[5]) for o in a[1:]), all(np.array_equal(d[0][5], o[5]) for o in d[1:])))
