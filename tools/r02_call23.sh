#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c23; mkdir -p $O
(time timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=6) > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -12 $O/pytest_gpu.log
