#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c14; mkdir -p $O
EK_TS_MAXN=1 timeout -k 10 300 python tools/two_stage_check.py 66 321 1000 1500 > $O/check.log 2>&1; echo "check rc=$?"; tail -2 $O/check.log
for e in 0 1 2; do echo "Q2 EXTRA=$e"; EK_Q2_EXTRA=$e timeout -k 10 300 python tools/two_stage_timing.py 16384 2>&1 | tail -1; done | tee $O/t.log
timeout -k 10 300 python tools/two_stage_timing.py 16384 1024 2>&1 | tail -1
timeout -k 10 300 python tools/two_stage_timing.py 8192 2>&1 | tail -1
timeout -k 10 300 python tools/two_stage_timing.py 4096 2>&1 | tail -1
