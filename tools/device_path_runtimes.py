#!/usr/bin/env python3
"""The device-resident solve (what bench.py's headline times) in a process WITHOUT PyTorch -- the system's HIP runtime --
and, with --torch, in one that imported and initialised PyTorch first (its bundled runtime then serves the library):
    python tools/device_path_runtimes.py [--torch]"""
import ctypes
import os
import sys
import time

if "--torch" in sys.argv:
    import torch
    torch.zeros(1, device="cuda"); torch.cuda.synchronize()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from eigenkernel_amd import solver  # noqa: E402

lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
dp = ctypes.POINTER(ctypes.c_double)


def alloc(nbytes):
    p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), nbytes) == 0; return p


for name, n, gep, nv, reps in (("c3", 16384, True, 16384, 3), ("c2", 4096, False, 4096, 10), ("c5", 16384, True, 1024, 3)):
    nn = n * n * 8
    dA, dZ, dw = alloc(nn), alloc(nn), alloc(n * 8)
    dB = alloc(nn) if gep else None
    st = np.zeros(8); tot = 0.0; stages = np.zeros(8)
    for r in range(reps + 1):
        assert lib.ek_hip_synth_matrix_device(n, 1, dA, n) == 0
        if gep: assert lib.ek_hip_synth_matrix_device(n, 2, dB, n) == 0
        lib.ek_hip_synchronize()
        t0 = time.perf_counter()
        info = lib.ek_hip_solve_device(1 if gep else 0, n, nv, dA, n, dB, n, dw, dZ, n, st.ctypes.data_as(dp), 8)
        lib.ek_hip_synchronize()
        dt = time.perf_counter() - t0
        assert info == 0
        if r > 0: tot += dt; stages += st
    print("%s: %.2f ms per solve (wall), stages %s" % (name, 1e3 * tot / reps, np.round(stages[:7] / reps, 4)), flush=True)
    for p in (dA, dZ, dw) + ((dB,) if gep else ()): lib.ek_hip_free(p)
with open("/proc/self/maps") as f:
    print(sorted({l.split()[-1] for l in f if "libamdhip64" in l}))
