# usage: tools/micro/ab_env.sh VAR   -> LM step time at configs[2], configs[1], configs[3] with VAR unset / set to 1, twice
for c in 2 1 3; do for rep in 1 2; do for v in "" 1; do
  if [ -n "$v" ]; then export $1=1; else unset $1; fi
  python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-parity-solver 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('config $c  $1=${v:-unset}', round(d['ms_per_step'],4), repr(d['config']['cost_last']))"
done; done; done
