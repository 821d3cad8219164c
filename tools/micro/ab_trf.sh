for v in head new head new; do MVUS_LIB_PATH=variants/libmvusba_$v.so python bench.py --solver trf --steps 5 --warmup 1 --no-cpu-baseline --no-parity-solver 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], repr(d['config']['cost_last']), d['kernels_ms']['jtu'])"; done
