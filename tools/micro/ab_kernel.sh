#!/bin/bash
# A/B of the residual+Jacobian kernel (rotating outputs) for library builds: tools/micro/ab_kernel.sh variants/libA.so variants/libB.so
for rep in 1 2 3; do
  for lib in "$@"; do
    MVUS_LIB_PATH=$PWD/$lib python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from mvus_amd import synth, problem as mp, ba
sc = synth.baseline_scene(2); prob, x0 = mp.problem_from_scene(sc)
h = ba.BAHandle(prob); h.set_x(x0)
h.time_kernel(1, 20)
print(os.environ['MVUS_LIB_PATH'].split('/')[-1], 'J kernel %.2f us, residual %.2f us, fused assembly %.1f us' % (1e3 * h.time_kernel(1, 100), 1e3 * h.time_kernel(0, 100), 1e3 * h.time_kernel(6, 20)))
PY
  done
done
