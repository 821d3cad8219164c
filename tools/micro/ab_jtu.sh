for v in "$@"; do MVUS_LIB_PATH=variants/libmvusba_$v.so python bench.py --solver trf --steps 5 --warmup 1 --no-cpu-baseline --no-parity-solver 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), repr(d['config']['cost_last']), round(1e3*d['kernels_ms']['jtu'],1))"; done
