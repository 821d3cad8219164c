#!/usr/bin/env python3
"""Assembly / Jacobian kernel time vs timeline density (detections per knot span): python tools/time_assembly.py <num_knots> ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvus_amd import ba, problem as mp, synth
for nk in [int(a) for a in sys.argv[1:]] or [5000]:
    kw = dict(synth.BASELINE_CONFIGS[2]); kw['num_knots'] = nk
    prob, x0 = mp.problem_from_scene(synth.make_scene(**kw))
    with ba.BAHandle(prob) as h:
        h.residual_jacobian(x0, ba.JAC_ANALYTIC)
        print('knots %d: %.1f detections per camera and span; assembly %.3f ms, J %.3f ms' % (
            nk, prob.M / prob.C / max(1, int(prob.n_coef.sum())), h.time_kernel(ba.KERNEL_ASSEMBLY, 20), h.time_kernel(ba.KERNEL_RESIDUAL_JACOBIAN, 20)))
