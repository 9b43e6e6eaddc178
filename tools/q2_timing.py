#!/usr/bin/env python3
"""Q2 application time by blocks of sweeps per pass: q2_timing.py [n ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eigenkernel_amd import solver  # noqa: E402

lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
sec = (ctypes.c_double * 4)()
flag = ctypes.c_int(0)
for n in [int(a) for a in sys.argv[1:]] or [4096, 16384]:
    for ncols in (n, min(n, 1024)):
        for nblk in (os.environ.get("Q2_TIMING_NBLK", "2,3,4").split(",")):
            os.environ["EK_Q2_NBLK"] = nblk
            lib.ek_hip_debug_two_stage_timing(n, ncols, 1, sec, ctypes.byref(flag))
            rc = lib.ek_hip_debug_two_stage_timing(n, ncols, 3, sec, ctypes.byref(flag))
            print("n=%5d columns %5d blocks per pass %2s rc=%d flag=%d  q2 %.4f s  (%.1f TFLOP/s algorithmic)" %
                  (n, ncols, nblk, rc, flag.value, sec[2], 2.0 * n * n * ncols / sec[2] / 1e12), flush=True)
