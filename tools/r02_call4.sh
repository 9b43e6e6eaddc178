#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c4; mkdir -p $O
EK_TS_MAXN=1 timeout -k 10 300 python tools/two_stage_check.py 5 66 130 321 700 1000 1500 > $O/check.log 2>&1; echo "check rc=$?"
tail -4 $O/check.log
for n in 4096 8192 16384; do timeout -k 10 300 python tools/two_stage_timing.py $n >> $O/t.log 2>&1; done; cat $O/t.log
EK_HIP_TWO_STAGE_MIN=100 timeout -k 10 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_blocks.py -x -q > $O/pytest_forced.log 2>&1; echo "forced rc=$?"; tail -5 $O/pytest_forced.log
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q --durations=5 > $O/pytest_configs.log 2>&1; echo "configs rc=$?"; tail -12 $O/pytest_configs.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-symv-events > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['stage_seconds_per_step'], d['parity'])"
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python3 tools/two_stage_timing.py 16384 16384 1 > $O/kt.log 2>&1; find /tmp/kt -name "*.db" | head -1 | xargs -r -I{} python tools/rocpd_summary.py {} > $O/kstats_16384.txt 2>&1; head -24 $O/kstats_16384.txt | cut -c1-180
