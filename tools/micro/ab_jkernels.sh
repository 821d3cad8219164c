#!/bin/bash
# A/B of the kernels that write / read the materialised Jacobian: tools/micro/ab_jkernels.sh libA.so libB.so ...
for rep in 1 2 3; do
  for lib in "$@"; do
    MVUS_LIB_PATH=$PWD/$lib python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from mvus_amd import synth, problem as mp, ba
sc = synth.baseline_scene(2); prob, x0 = mp.problem_from_scene(sc)
h = ba.BAHandle(prob); h.set_x(x0)
h.time_kernel(1, 20)
print(os.environ['MVUS_LIB_PATH'].split('/')[-1], 'J %.2f us (one buffer %.2f), Jv %.2f, JTu %.2f, assembly from J %.1f' % tuple(1e3 * h.time_kernel(w, n) for w, n in ((1, 100), (5, 20), (2, 50), (3, 50), (4, 20))))
PY
  done
done
