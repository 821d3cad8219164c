for c in 2 1; do for i in 1 2; do python bench.py --config $c --solver trf --steps 5 --warmup 1 --no-cpu-baseline --no-parity-solver 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config $c', d['ms_per_step'], repr(d['config']['cost_last']), d['kernels_ms']['jtu'])"; done; done
