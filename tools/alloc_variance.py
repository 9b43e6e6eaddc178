#!/usr/bin/env python3
"""Does the tridiagonalisation time depend on where the work arrays land in HBM?
Re-allocates the library's device memory (ek_hip_finalize) between N=16384 GEP solves, with
other allocations of varying size in between, and prints the stage time of each placement."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eigenkernel_amd import solver

n = 16384
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
dev = torch.device("cuda", 0)
ballast_gib = float(os.environ.get("BALLAST_GIB", "0"))
ballast = torch.empty((int(ballast_gib * (1 << 27)),), dtype=torch.float64, device=dev) if ballast_gib > 0 else None
dA = torch.empty((n, n), dtype=torch.float64, device=dev); dB = torch.empty_like(dA); dZ = torch.empty_like(dA)
dw = torch.empty((n,), dtype=torch.float64, device=dev)
stage = (ctypes.c_double * 8)()
hold = []
res = []
for rnd in range(rounds):
    for rep in range(2):
        lib.ek_hip_synth_matrix_device(n, 1, dA.data_ptr(), n); lib.ek_hip_synth_matrix_device(n, 2, dB.data_ptr(), n)
        assert lib.ek_hip_solve_device(1, n, n, dA.data_ptr(), n, dB.data_ptr(), n, dw.data_ptr(), dZ.data_ptr(), n, stage, 8) == 0
    res.append(round(stage[2], 4))
    lib.ek_hip_finalize()
    hold.append(torch.empty(((rnd * 7 % 5 + 1) << 26,), dtype=torch.float64, device=dev))   # 0.5 .. 2.5 GiB
print("ballast %.0f GiB sytrd by placement:" % ballast_gib, res, flush=True)
