import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from golden_util import load_case
from mvus_amd import _lib, problem as mp
from mvus_amd.ba import BAHandle
for sync in (True, False):
    scene, g = load_case('rs_F_2int_3cam'); scene.settings['opt_sync'] = sync
    prob, x0 = mp.problem_from_scene(scene)
    with BAHandle(prob) as h:
        o = _lib.default_opts(_lib.SOLVER_LM_SCHUR, _lib.JAC_ANALYTIC, 6); o.verbose = 2
        r = h.solve(x0, opts=o)
        print('sync', sync, 'materialize', os.environ.get('MVUS_LM_MATERIALIZE_J'), r.initial_cost, r.cost, r.nfev, r.njev, r.status, r.lin_iters)
        f, J, ctrl = h.residual_jacobian(x0, _lib.JAC_ANALYTIC)
        gg, A, band, cross = h.normal_equations()
        print('  A[0] diag', np.diag(A[0])[:4], 'g[:6]', gg[:6])
        p = h.lm_step(1e-4)
        print('  lm_step p[:9]', p[:9], 'finite', np.isfinite(p).all(), 'max|p|', np.abs(p).max())
