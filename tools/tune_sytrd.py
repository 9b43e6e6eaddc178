#!/usr/bin/env python3
"""Micro-benchmark of the tridiagonalisation's first panels (EK_SYTRD_MAXCOLS) for tuning:
prints stage seconds, and the symv event profile (GB/s) for a given n / ld / G."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ld = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
sec = ctypes.c_double(0)
assert lib.ek_hip_debug_sytrd(n, ld, 1, ctypes.byref(sec)) == 0      # warm-up
lib.ek_hip_profile_symv(1)
assert lib.ek_hip_debug_sytrd(n, ld, reps, ctypes.byref(sec)) == 0
s, l, b = ctypes.c_double(0), ctypes.c_longlong(0), ctypes.c_double(0)
lib.ek_hip_profile_symv_get(ctypes.byref(s), ctypes.byref(l), ctypes.byref(b))
lib.ek_hip_profile_symv(0)
print("n=%d ld=%d maxcols=%s G=%s: stage %.4f s/rep; symv %d launches avg %.2f us, %.1f GB/s; non-symv %.2f us/col"
      % (n, ld, os.environ.get("EK_SYTRD_MAXCOLS"), os.environ.get("EK_SYMV_G"), sec.value, l.value,
         1e6 * s.value / max(l.value, 1), b.value / max(s.value, 1e-30) / 1e9,
         1e6 * (sec.value * reps - s.value) / max(l.value, 1)))
