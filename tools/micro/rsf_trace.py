"""rs_F_2int_3cam, the reference's second BA from the reference's own start with the reference's own matrix: the solution after
k = 1, 2, ... evaluations (max_nfev = k; every run starts again from x0, all three are deterministic) by
    scipy   the oracle's least_squares call (scipy's trf / lsmr / 2-point differences on the numpy residual: the reference's algorithm
            on the reference's arithmetic, to 4.5e-13 px)
    host    this library's restatement (ba_solver.h) over the host build of the device math (tests/hostcheck; sequential sums)
    gpu     the same restatement on the GPU (tree sums, fused multiply-adds)          [only when a GPU is present]
and where they part: max |x_a - x_b| scaled by max(1, |x|), and the cost difference, per k."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from test_fd_mode_host import filtered_case, golden_matrix
from hostcheck_util import HostHandle
from mvus_amd import _lib, problem as mp
from oracle import ba_oracle as orc

name = sys.argv[1] if len(sys.argv) > 1 else 'rs_F_2int_3cam'
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scene, g = filtered_case(name)
prob, _ = mp.problem_from_scene(scene); oprob, _ = orc.problem_from_scene(scene)
A = golden_matrix(g, second=True)
x0 = g['ba2_200_x0']
gpu = None
try:
    import torch
    if torch.cuda.is_available():
        from mvus_amd.ba import BAHandle
        gpu = BAHandle(prob)
except Exception as e:
    print('no GPU:', e)
host = HostHandle(prob)
sc = np.maximum(1.0, np.abs(x0))
print('# %s: n = %d, m = %d; x differences are max |dx| / max(1, |x0|)' % (name, x0.size, prob.n_residuals))
print('%3s | %-22s %-22s %-22s | %-12s %-12s %-12s | nfev/status scipy host gpu' % ('k', 'cost scipy', 'cost host', 'cost gpu', 'host-scipy', 'gpu-scipy', 'gpu-host'))
for k in range(1, kmax + 1):
    rs = orc.solve(oprob, x0, max_iter=k, pattern=A)
    xh, rh, _ = host.solve(x0, _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_FD, k), matrix=A)
    if gpu is not None:
        rg = gpu.solve(x0, solver=_lib.SOLVER_TRF_LSMR, jac_mode=_lib.JAC_FD, max_nfev=k, matrix=A)
        xg, cg, ng = rg.x, rg.cost, (rg.nfev, rg.status)
    else:
        xg, cg, ng = None, float('nan'), None
    d = lambda a, b: float(np.max(np.abs(a - b) / sc)) if (a is not None and b is not None) else float('nan')
    print('%3d | %-22.15e %-22.15e %-22.15e | %-12.3e %-12.3e %-12.3e | %s %s %s' % (k, rs.cost, rh.cost, cg, d(xh, rs.x), d(xg, rs.x), d(xg, xh),
          (rs.nfev, rs.status), (rh.nfev, rh.status), ng), flush=True)
