#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c7; mkdir -p $O
EK_TS_MAXN=1 timeout -k 10 300 python tools/two_stage_check.py 5 66 130 321 700 1000 1500 > $O/check.log 2>&1; echo "check rc=$?"; grep -c "e-1[3-9]\|e-0" $O/check.log; tail -3 $O/check.log
for n in 4096 8192 16384; do timeout -k 10 300 python tools/two_stage_timing.py $n 2>&1 | tail -1; done | tee $O/t.log
