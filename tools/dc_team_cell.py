#!/usr/bin/env python3
"""One grid cell of a 1 x P team solved on this GPU with the divide & conquer's team form rehearsed (tools, not product).

  python tools/dc_team_cell.py N P [levels] [reps]

Prints the cell's stage seconds and the D&C's per-rank figure; run under `rocprofv3 --kernel-trace --stats` for the
kernels of the stage (dc_*, the merge GEMMs)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver  # noqa: E402

n, P = int(sys.argv[1]), int(sys.argv[2])
levels = int(sys.argv[3]) if len(sys.argv) > 3 else -1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
lib = solver.load_library()
assert lib.ek_hip_init(0) == 0


def dmalloc(nbytes):
    ptr = ctypes.c_void_p()
    assert lib.ek_hip_malloc(ctypes.byref(ptr), int(nbytes)) == 0
    return ptr


dA, dB, dw = dmalloc(n * n * 8), dmalloc(n * n * 8), dmalloc(n * 8)
dZ = dmalloc(((n + P - 1) // P + 64) * n * 8)
stage = (ctypes.c_double * 8)()
dc = (ctypes.c_double * 3)()
for rep in range(reps):
    assert lib.ek_hip_synth_matrix_device(n, 1, dA, n) == 0
    assert lib.ek_hip_synth_matrix_device(n, 2, dB, n) == 0
    assert lib.ek_hip_debug_stedc_team(P if levels != 0 else 0, levels, 1) == 0
    rc = lib.ek_hip_solve_device_grid(1, n, n, dA, n, dB, n, dw, dZ, n, 64, 1, P, 0, 0, stage, 8)
    assert rc == 0, rc
    assert lib.ek_hip_debug_stedc_team_get(dc) == 0
    assert lib.ek_hip_debug_stedc_team(0, -1, 0) == 0
    print("n=%d P=%d levels=%d: stedc stage %.4f s; sections %.4f, longest rank's %.4f -> %.4f s per rank; Q2+Q1 %.4f, recovery %.4f"
          % (n, P, levels, stage[4], dc[1], dc[2], stage[4] - dc[1] + dc[2], stage[5], stage[6]), flush=True)
