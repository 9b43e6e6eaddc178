#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c9; mkdir -p $O
EK_TS_MAXN=1 timeout -k 10 300 python tools/two_stage_check.py 5 66 130 321 700 1000 1500 2500 > $O/check.log 2>&1; echo "check rc=$?"; tail -3 $O/check.log
for n in 4096 8192 16384; do EK_SY2SB_PROF=1 timeout -k 10 300 python tools/two_stage_timing.py $n 2>&1 | tail -2; done | tee $O/t.log
EK_SY2SB_LOOKAHEAD_MIN=100000000 timeout -k 10 300 python tools/two_stage_timing.py 16384 2>&1 | tail -1
EK_HIP_TWO_STAGE_MIN=100 timeout -k 10 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_blocks.py -x -q 2>&1 | tail -3
