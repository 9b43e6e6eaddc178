#!/usr/bin/env python3
"""Does a short probe (the first panels of the tridiagonalisation) predict which of the two
placement modes ('run-to-run spread, explained') a workspace allocation is in?
For several placements of the library's workspace: probe time (first 128 columns, 5 reps) and the
full N=16384 tridiagonalisation time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver
n = 16384
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
sec = ctypes.c_double(0)
hold = []
for rnd in range(rounds):
    lib.ek_hip_debug_set_sytrd_maxcols(-1)
    lib.ek_hip_debug_sytrd(n, 0, 1, ctypes.byref(sec))          # allocates the workspace, warms up
    lib.ek_hip_debug_set_sytrd_maxcols(128)
    probes = []
    for _ in range(3):
        lib.ek_hip_debug_sytrd(n, 0, 5, ctypes.byref(sec)); probes.append(sec.value * 1e3)
    lib.ek_hip_debug_set_sytrd_maxcols(-1)
    lib.ek_hip_debug_sytrd(n, 0, 2, ctypes.byref(sec)); full = sec.value
    print("placement %d: probe (128 columns) %s ms, full sytrd %.4f s" % (rnd, ["%.3f" % p for p in probes], full), flush=True)
    lib.ek_hip_finalize()
    p = ctypes.c_void_p()
    assert lib.ek_hip_malloc(ctypes.byref(p), ((rnd * 7 % 5 + 1) << 29)) == 0     # 0.5 .. 2.5 GiB ballast, kept
    hold.append(p)
