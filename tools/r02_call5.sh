#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c5; mkdir -p $O
for e in 0 1 2 3; do echo "EXTRA=$e"; EK_SB2ST_EXTRA=$e timeout -k 10 200 python tools/two_stage_timing.py 8192 64 2; done 2>&1 | tee $O/extra.log
echo "WGS=1 n=2048 (pure task rate)"; EK_SB2ST_WGS=1 timeout -k 10 200 python tools/two_stage_timing.py 2048 64 1 2>&1 | tee -a $O/extra.log
EK_SB2ST_EXTRA=1 EK_TS_MAXN=1 timeout -k 10 300 python tools/two_stage_check.py 321 700 1000 1500 2>&1 | tail -3
