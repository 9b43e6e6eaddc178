#!/usr/bin/env python3
"""One pass of the two-stage pieces on the synthetic matrix (for rocprofv3 counter passes): two_stage_once.py [n] [ncols]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eigenkernel_amd import solver  # noqa: E402

lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else n
sec = (ctypes.c_double * 4)(); flag = ctypes.c_int(0)
rc = lib.ek_hip_debug_two_stage_timing(n, ncols, 1, sec, ctypes.byref(flag))
print("rc", rc, "flag", flag.value, list(sec))
