#!/bin/bash
# A/B of two library builds on one box: tools/micro/ab_bench.sh variants/libmvusba_old.so variants/libmvusba_new.so
for rep in 1 2 3; do
  for lib in "$@"; do
    MVUS_LIB_PATH=$PWD/$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity-solver 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '%.4f ms/step' % d['ms_per_step'])"
  done
done
