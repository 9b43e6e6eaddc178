#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c21; mkdir -p $O
for n in 6144 8192 10240; do for m in 0 1; do echo "n=$n two_stage_min=$m"; EK_HIP_TWO_STAGE_MIN=$m python bench.py --order $n --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-symv-events | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), {k.split(':')[-1]:round(v,4) for k,v in d['stage_seconds_per_step'].items() if v>1e-3})"; done; done 2>&1 | tee $O/crossover.log
for n in 8192; do for m in 0 1; do echo "SEP n=$n two_stage_min=$m"; EK_HIP_TWO_STAGE_MIN=$m python bench.py --order $n --problem sep --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-symv-events | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1))"; done; done 2>&1 | tee -a $O/crossover.log
