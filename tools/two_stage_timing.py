#!/usr/bin/env python3
"""Timing of the two-stage pieces on synthetic matrices: two_stage_timing.py n [ncols] [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver
lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
sec = (ctypes.c_double * 4)(); flag = ctypes.c_int(0)
n = int(sys.argv[1]); ncols = int(sys.argv[2]) if len(sys.argv) > 2 else n; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib.ek_hip_debug_two_stage_timing(n, ncols, 1, sec, ctypes.byref(flag))
rc = lib.ek_hip_debug_two_stage_timing(n, ncols, reps, sec, ctypes.byref(flag))
print("timing n=%5d ncols=%d rc=%d flag=%d  sy2sb %.4f s  sb2st %.4f s  q2 %.4f s  q1 %.4f s  total %.4f s"
      % (n, ncols, rc, flag.value, sec[0], sec[1], sec[2], sec[3], sum(sec)), flush=True)
