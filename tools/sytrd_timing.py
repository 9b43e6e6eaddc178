#!/usr/bin/env python3
"""Stage time of the single-GPU tridiagonalisation (tools): python tools/sytrd_timing.py N [N ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver
lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
sec = ctypes.c_double(0)
for n in [int(x) for x in sys.argv[1:]] or [16384]:
    lib.ek_hip_debug_sytrd(n, 0, 1, ctypes.byref(sec))
    lib.ek_hip_debug_sytrd(n, 0, 2, ctypes.byref(sec))
    print("n=%d sytrd %.4f s = %.2f us per column" % (n, sec.value, 1e6 * sec.value / n), flush=True)
