# usage: tools/micro/ab_lm.sh variant...   -> LM step time and fused-assembly kernel time per variants/libmvusba_<variant>.so (configs[2])
for v in "$@"; do MVUS_LIB_PATH=variants/libmvusba_$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-parity-solver 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), repr(d['config']['cost_last']), round(1e3*d['kernels_ms']['fused_jacobian_normal_eq_assembly'],1))"; done
