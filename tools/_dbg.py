import numpy as np, sys
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from golden_util import load_case
from mvus_amd import _lib, problem as mp
from mvus_amd.ba import BAHandle
from hostcheck_util import HostHandle
scene,g=load_case('c1_pinhole_2cam')
prob,_=mp.problem_from_scene(scene)
opts=_lib.default_opts(_lib.SOLVER_LM_SCHUR,_lib.JAC_ANALYTIC,3); opts.verbose=2
xh,rh,fh=HostHandle(prob).solve(g['x0'],opts)
with BAHandle(prob) as h:
    r=h.solve(g['x0'],opts=opts)
print('host',rh.cost,rh.nfev,'gpu',r.cost,r.nfev, np.abs(r.x-xh).max())
