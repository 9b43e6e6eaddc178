#!/usr/bin/env python3
"""Stage times of the two-stage pieces for both bulge-chasing kernels: chase_timing.py [n ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eigenkernel_amd import solver  # noqa: E402

lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
sec = (ctypes.c_double * 4)()
flag = ctypes.c_int(0)
for n in [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192, 16384]:
    for mode in (1, 2):
        os.environ["EK_SB2ST_CHASE"] = str(mode)
        lib.ek_hip_debug_two_stage_timing(n, min(n, 1024), 1, sec, ctypes.byref(flag))
        rc = lib.ek_hip_debug_two_stage_timing(n, min(n, 1024), 3, sec, ctypes.byref(flag))
        print("n=%5d chase mode %d rc=%d flag=%d  sy2sb %.4f s  sb2st %.4f s  q2 (%d columns) %.4f s" %
              (n, mode, rc, flag.value, sec[0], sec[1], min(n, 1024), sec[2]), flush=True)
