"""rs_F_2int_3cam, first accepted step of the reference's second BA (5 evaluations: four shrinking trials, then the accepted one), with
LSMR capped at k iterations: scipy's least_squares against this library's restatement, BOTH on the same residual function (the host
build of the device math) -- f, the finite-difference Jacobian and the column groups are bit-identical at the start
(tools/micro/rsf_trace.py checks it).  Uncapped, LSMR stops at its iteration limit min(m, n) = 177 in this step (lsmr_itn=177 in the
verbose trace): it has not converged, and what it returns carries the rounding history of 177 iterations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from scipy.optimize import least_squares
from test_fd_mode_host import filtered_case, golden_matrix
from hostcheck_util import HostHandle
from mvus_amd import _lib, problem as mp
from oracle import ba_oracle as orc
scene, g = filtered_case('rs_F_2int_3cam')
prob, _ = mp.problem_from_scene(scene); oprob, _ = orc.problem_from_scene(scene)
A = golden_matrix(g, second=True); x0 = g['ba2_200_x0']
host = HostHandle(prob)
sc = np.maximum(1.0, np.abs(x0))
print('# LSMR cap | cost after the first accepted step: scipy, restatement | max |x_scipy - x_restatement| / max(1, |x0|)')
for k in (2, 3, 5, 8, 12, 20, 30, 40, 50, 80, 120, 177):
    r = least_squares(lambda x: host.residual(x), x0, jac_sparsity=A, tr_solver='lsmr', tr_options=dict(maxiter=k), xtol=1e-12, max_nfev=5, bounds=orc.bounds(oprob))
    o = _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_FD, 5); o.lsmr_maxiter = k
    xh, rh, _ = host.solve(x0, o, matrix=A)
    print('%4d | %.12e %.12e | %.2e' % (k, r.cost, rh.cost, np.max(np.abs(xh - r.x) / sc)), flush=True)
