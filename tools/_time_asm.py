import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvus_amd import ba, problem as mp, synth
sc = synth.baseline_scene(2)
prob, x0 = mp.problem_from_scene(sc)
with ba.BAHandle(prob) as h:
    h.residual_jacobian(x0, ba.JAC_ANALYTIC)
    print('assembly ms', h.time_kernel(ba.KERNEL_ASSEMBLY, 20))
