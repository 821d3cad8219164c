"""Fresh handle + eight-evaluation LM solve, many times, at configs[1] and configs[4] (and [2] with --all): the distinct outcomes.
With MVUS_DET_ASSEMBLY=1 (or --det) there must be exactly ONE outcome per configuration, to the last bit of cost and x."""
import os, sys, collections, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mvus_amd import synth, problem as mp, ba
args = [a for a in sys.argv[1:] if not a.startswith('--')]
det = '--det' in sys.argv
reps = int(args[0]) if args else 60
for index in ((1, 4, 2) if '--all' in sys.argv else (1, 4)):
    sc = synth.baseline_scene(index); prob, x0 = mp.problem_from_scene(sc)
    outcomes = collections.Counter()
    for rep in range(reps):
        with ba.BAHandle(prob) as h:
            if det:
                h.set_deterministic(True)
            r = h.solve(x0, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=8)
            outcomes[(repr(r.cost), hashlib.sha1(np.ascontiguousarray(r.x).tobytes()).hexdigest()[:12], r.nfev, r.njev, r.status, r.cost < r.initial_cost)] += 1
    print('config', index, 'deterministic' if det or os.environ.get('MVUS_DET_ASSEMBLY') else 'atomic', '-- distinct (cost, sha1(x), nfev, njev, status, descended):', len(outcomes))
    for k, v in outcomes.most_common(4):
        print('   ', v, 'x', k)
