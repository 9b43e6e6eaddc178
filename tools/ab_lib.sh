#!/bin/sh
# Same-box A/B of two library builds: tools/ab_lib.sh libA.so libB.so [bench args]
a=$1; b=$2; shift 2
for rep in 1 2 3; do
  for l in $a $b; do
    EK_HIP_LIB=$l python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-check --no-symv-events "$@" |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_seconds_per_step']; print('$l', round(d['ms_per_step'],1), {k.split(':')[-1]: round(x,4) for k,x in s.items() if x > 1e-4})"
  done
done
