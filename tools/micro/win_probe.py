"""Cycle-counter probe build of the window-major assembly (variants/libmvusba_probe.so, -DMVUS_WIN_PROBE=1): one fused assembly, the
kernel prints per-phase cycles (s_memtime, 100 MHz -> x24 for shader cycles at 2.4 GHz) of a few wavefronts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvus_amd import ba, problem as mp, synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prob, x0 = mp.problem_from_scene(synth.make_scene(**dict(synth.BASELINE_CONFIGS[cfg])))
with ba.BAHandle(prob) as h:
    h.residual_jacobian(x0)
    h.normal_equations()
