"""Where do the converged second BAs of rs_F_2int_3cam land, by WHO solves and WHOSE residual arithmetic is used?  12 starts each (the
reference's start, and 11 copies perturbed by 1e-15 relative), max_iter = 200, the reference's matrix:
    E1  scipy least_squares  + the oracle's numpy residual      (the reference's algorithm and arithmetic to 4.5e-13 px)
    E2  scipy least_squares  + the host build of the device math as `fun` (same solver, the library's residual arithmetic)
    E3  the library's restatement (ba_solver.h) + the host build  (tests/hostcheck)
    E4  the library on the GPU (when present)
printed: final RMSE minus the reference's ba2_200_rmse, sorted."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from scipy.optimize import least_squares
from test_fd_mode_host import filtered_case, golden_matrix
from hostcheck_util import HostHandle
from mvus_amd import _lib, problem as mp
from oracle import ba_oracle as orc
name = sys.argv[1] if len(sys.argv) > 1 else 'rs_F_2int_3cam'
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 12
scene, g = filtered_case(name)
prob, _ = mp.problem_from_scene(scene); oprob, _ = orc.problem_from_scene(scene)
A = golden_matrix(g, second=True)
ref_rmse = float(g['ba2_200_rmse'])
host = HostHandle(prob)
gpu = None
try:
    import torch
    if torch.cuda.is_available():
        from mvus_amd.ba import BAHandle
        gpu = BAHandle(prob)
except Exception:
    pass
starts = []
for k in range(nm):
    rng = np.random.default_rng(500 + k)
    starts.append(g['ba2_200_x0'] * (1.0 + (1e-15 * rng.standard_normal(g['ba2_200_x0'].size) if k else 0.0)))
def rmse(x): return orc.reprojection_rmse(oprob, x) - ref_rmse
def scipy_run(fun, x0): return least_squares(fun, x0, jac_sparsity=A, tr_solver='lsmr', xtol=1e-12, max_nfev=200, bounds=orc.bounds(oprob))
out = {}
only_e3 = os.environ.get('RSF_ONLY_E3') is not None
if not only_e3:
  out['E1 scipy + numpy residual'] = [(rmse(r.x), r.nfev, r.status) for r in (scipy_run(lambda x: orc.residual(oprob, x), x0) for x0 in starts)]
if not only_e3:
  out['E2 scipy + host-build residual'] = [(rmse(r.x), r.nfev, r.status) for r in (scipy_run(lambda x: host.residual(x), x0) for x0 in starts)]
e3 = []
for x0 in starts:
    x, res, _ = host.solve(x0, _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_FD, 200), matrix=A)
    e3.append((rmse(x), res.nfev, res.status))
out['E3 restatement + host-build residual' + (' (long-double sums)' if os.environ.get('MVUS_HOST_EXACT_SUMS') else '')] = e3
if gpu is not None:
    out['E4 GPU'] = [(lambda r: (rmse(r.x), r.nfev, r.status))(gpu.solve(x0, solver=_lib.SOLVER_TRF_LSMR, jac_mode=_lib.JAC_FD, max_nfev=200, matrix=A)) for x0 in starts]
print('# %s: final RMSE - reference (px), %d starts (first = unperturbed), sorted; nfev range; statuses' % (name, nm))
for k, v in out.items():
    d = np.array([a[0] for a in v])
    print('%-40s unperturbed %+.2e | min %+.2e median %+.2e max %+.2e | nfev %d..%d | status %s' % (k, d[0], d.min(), np.median(d), d.max(), min(a[1] for a in v), max(a[1] for a in v), sorted(set(a[2] for a in v))))
    print('    ' + ' '.join('%+.1e' % t for t in np.sort(d)))
