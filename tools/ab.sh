#!/bin/sh
# Same-box A/B of an environment knob: tools/ab.sh VAR "v1 v2 ..." [bench args]
# (boxes of the pool differ by up to ~20 %: only compare numbers from ONE gpurun call)
var=$1; vals=$2; shift 2
for rep in 1 2; do
  for v in $vals; do
    env "$var=$v" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check --no-symv-events "$@" |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_seconds_per_step']; print('$var=$v', round(d['ms_per_step'],1), {k.split(':')[-1]: round(x,4) for k,x in s.items() if x > 1e-4})"
  done
done
