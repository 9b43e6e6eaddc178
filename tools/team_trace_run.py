#!/usr/bin/env python3
"""One rehearsal of the team form of the dense -> band stage (whole team on this GPU, look-ahead on) for a kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 tools/team_trace_run.py [n] [P]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
sec = ctypes.c_double(0)
assert lib.ek_hip_debug_sy2sb_team_profile(n, P, 1, -1, ctypes.byref(sec), None) == 0
print("n=%d team of %d rehearsed with look-ahead: %.4f s" % (n, P, sec.value))
