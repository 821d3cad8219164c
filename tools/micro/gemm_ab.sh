# usage (GPU box): tools/micro/gemm_ab.sh variant... -> LM step time per variants/libmvusba_<variant>.so at configs[2] and [3], default slab plan and 8 / 16 slabs
run() { python bench.py --config $2 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-solver 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 config $2 slabs ${3:-plan}', round(d['ms_per_step'],4), repr(d['config'].get('cost_last')))"; }
for v in "$@"; do export MVUS_LIB_PATH=variants/libmvusba_$v.so; for c in 2 3; do
  unset MVUS_GEMM_SLABS; run $v $c
  for s in 8 16; do export MVUS_GEMM_SLABS=$s; run $v $c $s; done
done; done
