"""Stress: many fresh handles, LM solves on motion-regularised scenes (one rank, and a one-rank handle with an identity all-reduce),
every outcome must equal the first one bit for bit; prints what differs."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np
from golden_util import load_case
from mvus_amd import ba, _lib, problem as mp, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cases = []
for name in ('rs_F_2int_3cam', 'calib_KE_bounds_3cam'):
    scene, g = load_case(name); prob, _ = mp.problem_from_scene(scene); cases.append((name, prob, g['x0']))
sc = synth.make_scene(3, 5000, seed=53, rolling_shutter=True, num_knots=260, motion_reg=True, motion_type='F', motion_weights=30.0)
prob, x0 = mp.problem_from_scene(sc); cases.append(('synth_3cam_F', prob, x0))
bad = 0
for name, prob, x0 in cases:
    ref = {}
    for i in range(reps):
        for mode in ('plain', 'identity'):
            with ba.BAHandle(prob) as h:
                if mode == 'identity': h.set_allreduce(lambda p, c, s: None, is_root=True)
                r = h.solve(x0, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=6)
                h.residual_jacobian(x0); p = h.lm_step(0.3)
                key = (repr(r.cost), r.nfev, hashlib.sha1(p.tobytes()).hexdigest()[:10])
                if mode not in ref: ref[mode] = key
                elif key != ref[mode]:
                    bad += 1; print('MISMATCH', name, mode, i, key, 'vs', ref[mode], flush=True)
    print(name, ref, flush=True)
print('mismatches:', bad)
