import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from test_fd_mode_host import filtered_case, golden_matrix
from golden_util import CASES
from mvus_amd import _lib, problem as mp
from mvus_amd.ba import BAHandle
from oracle import ba_oracle as orc
for name in CASES:
    scene, g = filtered_case(name)
    prob,_ = mp.problem_from_scene(scene); oprob,_ = orc.problem_from_scene(scene)
    ds=[]
    for rep in range(6):
        with BAHandle(prob) as h:
            r = h.solve(g['ba2_200_x0'], solver=_lib.SOLVER_TRF_LSMR, jac_mode=_lib.JAC_FD, max_nfev=200, matrix=golden_matrix(g, second=True))
        ds.append(orc.reprojection_rmse(oprob, r.x)-float(g['ba2_200_rmse']))
    print(name, ' '.join('%+.1e'%d for d in ds))
