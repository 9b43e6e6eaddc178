#!/bin/bash
export TMPDIR=/tmp
for w in 4 8 16; do echo "WAVES=$w"; EK_SB2ST_WAVES=$w EK_TS_MAXN=1 timeout -k 10 200 python tools/two_stage_check.py 321 1000 2>&1 | tail -1; EK_SB2ST_WAVES=$w timeout -k 10 200 python tools/two_stage_timing.py 8192 64 2>&1 | tail -1; done
