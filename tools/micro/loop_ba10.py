import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from test_fd_mode_host import golden_matrix
from golden_util import CASES, load_case
from mvus_amd import _lib, problem as mp
from mvus_amd.ba import BAHandle
from oracle import ba_oracle as orc
for name in CASES:
    scene, g = load_case(name)
    prob,_ = mp.problem_from_scene(scene); oprob,_ = orc.problem_from_scene(scene)
    out=[]
    for rep in range(5):
        with BAHandle(prob) as h:
            r = h.solve(g['x0'], solver=_lib.SOLVER_TRF_LSMR, jac_mode=_lib.JAC_PATTERN, max_nfev=10, matrix=golden_matrix(g))
            keep = h.outlier_mask(r.x, float(g['thres_outlier']))
        out.append('cost %+.1e rmse %+.1e agree %.3f' % (r.cost/float(g['ba10_cost'])-1, orc.reprojection_rmse(oprob, r.x)-float(g['ba10_rmse']), np.mean(keep.astype(np.uint8)==g['outlier_keep'])))
    print(name, ' | '.join(out))
