#!/usr/bin/env python3
"""ek_hip_solve on host arrays at a BASELINE order, several times in one process, with the staging pipeline's own trace
(EK_HIP_PIPE_TRACE=1: rates per direction, what the main thread waited for):
    EK_HIP_PIPE_TRACE=1 python tools/host_path_trace.py [n] [reps] [problem]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from eigenkernel_amd import solver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
problem = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
for r in range(reps):
    out = bench.host_path_step(lib, solver, problem, n, n)
    print("rep %d: %s" % (r, out), flush=True)
