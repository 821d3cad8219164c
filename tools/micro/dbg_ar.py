import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from golden_util import load_case
from mvus_amd import ba, _lib, problem as mp
scene, g = load_case('rs_F_2int_3cam')
prob, x0 = mp.problem_from_scene(scene)
opts = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 6); opts.verbose = 2
for mode in ('plain', 'identity-callback', 'identity-callback-atomic'):
    if mode.endswith('atomic'): os.environ['MVUS_ASM_ATOMIC'] = '1'
    with ba.BAHandle(prob) as h:
        if mode != 'plain': h.set_allreduce(lambda p, c, s: None, is_root=True)
        print('==', mode, flush=True)
        r = h.solve(g['x0'], opts=opts)
        print(mode, repr(r.cost), r.nfev, flush=True)
