#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c13; mkdir -p $O
EK_TS_MAXN=1 timeout -k 10 300 python tools/two_stage_check.py 5 66 130 321 700 1000 1500 > $O/check.log 2>&1; echo "check rc=$?"; grep -c "flag=0" $O/check.log; tail -2 $O/check.log
for e in 0 2 4; do echo "Q2 EXTRA=$e"; EK_Q2_EXTRA=$e timeout -k 10 300 python tools/two_stage_timing.py 16384 2>&1 | tail -1; EK_Q2_EXTRA=$e timeout -k 10 300 python tools/two_stage_timing.py 16384 1024 2>&1 | tail -1; done | tee $O/t.log
EK_Q2_WGS=256 timeout -k 10 300 python tools/two_stage_timing.py 16384 2>&1 | tail -1
timeout -k 10 300 python tools/two_stage_timing.py 8192 2>&1 | tail -1
