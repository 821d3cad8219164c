#!/usr/bin/env python3
"""Time the hot kernels of one workload with the library given by MVUS_LIB_PATH (kernel experiments)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvus_amd import ba, problem as mp, synth

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sc = synth.baseline_scene(cfg)
prob, x0 = mp.problem_from_scene(sc)
with ba.BAHandle(prob) as h:
    h.set_x(x0)
    out = {'lib': os.environ.get('MVUS_LIB_PATH', 'default')}
    for name, k in (('residual_jacobian', 1), ('jv', 2), ('jtu', 3), ('assembly', 4)):
        out[name] = round(h.time_kernel(k, 20) * 1e3, 1)
    print(json.dumps(out))
