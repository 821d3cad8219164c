"""How the GPU's reference-algorithm mode (TRF + LSMR + grouped 2-point differences) scatters when ITS start is perturbed in the last
place -- the counterpart of tests/golden/ens_*.npz (the real reference under last-place noise on its residuals)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import gauge
from golden_util import reference_spread
from test_fd_mode_host import filtered_case, golden_matrix
from mvus_amd import _lib, problem as mp
from mvus_amd.ba import BAHandle
from oracle import ba_oracle as orc
for name in sys.argv[1:] or ['c1_pinhole_2cam', 'rs_F_2int_3cam', 'dist_fixed_2cam', 'config1_shape_7cam']:
    scene, g = filtered_case(name)
    prob, _ = mp.problem_from_scene(scene); oprob, _ = orc.problem_from_scene(scene)
    spread = reference_spread(oprob, name, g['ba2_200_x'])
    rows = []
    with BAHandle(prob) as h:
        for k in range(12):
            rng = np.random.default_rng(500 + k)
            x0 = g['ba2_200_x0'] * (1.0 + (1e-15 * rng.standard_normal(g['ba2_200_x0'].size) if k else 0.0))
            r = h.solve(x0, solver=_lib.SOLVER_TRF_LSMR, jac_mode=_lib.JAC_FD, max_nfev=200, matrix=golden_matrix(g, second=True))
            c = gauge.compare(oprob, g['ba2_200_x'], r.x)
            rows.append((c['rmse_b'] - float(g['ba2_200_rmse']), c['traj_rms'], c['dbeta_max'], r.nfev, r.status))
    d = np.array([r[0] for r in rows]); t = np.array([r[1] for r in rows]); b = np.array([r[2] for r in rows])
    print('%-22s GPU, 12 starts (first unperturbed): RMSE - ref %+.1e .. %+.1e (unperturbed %+.1e), traj rms vs ref %.1e .. %.1e, dbeta %.1e .. %.1e, nfev %s | reference ensemble: rmse %.1e traj %.1e dbeta %.1e'
          % (name, d.min(), d.max(), d[0], t.min(), t.max(), b.min(), b.max(), sorted(set(r[3] for r in rows)), spread['rmse'], spread['traj_rms'], spread['dbeta_max']), flush=True)
