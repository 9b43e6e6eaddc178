#!/usr/bin/env python3
"""Effect of the placement-aware workspace (EK_HIP_PLACEMENT) on the N=16384 GEP solve: several fresh
allocations of the workspace in one process, stage time of the tridiagonalisation of each."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eigenkernel_amd import solver
n = 16384
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
dev = torch.device("cuda", 0)
dA = torch.empty((n, n), dtype=torch.float64, device=dev); dB = torch.empty_like(dA); dZ = torch.empty_like(dA)
dw = torch.empty((n,), dtype=torch.float64, device=dev)
stage = (ctypes.c_double * 8)()
hold, res, tot = [], [], []
for rnd in range(rounds):
    for rep in range(2):
        lib.ek_hip_synth_matrix_device(n, 1, dA.data_ptr(), n); lib.ek_hip_synth_matrix_device(n, 2, dB.data_ptr(), n)
        assert lib.ek_hip_solve_device(1, n, n, dA.data_ptr(), n, dB.data_ptr(), n, dw.data_ptr(), dZ.data_ptr(), n, stage, 8) == 0
    res.append(round(stage[2], 4)); tot.append(round(sum(stage[i] for i in range(7)), 4))
    lib.ek_hip_finalize()
    hold.append(torch.empty(((rnd * 7 % 5 + 1) << 26,), dtype=torch.float64, device=dev))
print("EK_HIP_PLACEMENT=%s sytrd by allocation: %s  solve: %s" % (os.environ.get("EK_HIP_PLACEMENT", "1"), res, tot), flush=True)
