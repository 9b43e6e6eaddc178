import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
from eigenkernel_amd import solver
n = 16384
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
dev = torch.device("cuda", 0)
dA = torch.empty((n, n), dtype=torch.float64, device=dev); dB = torch.empty_like(dA); dZ = torch.empty_like(dA)
dw = torch.empty((n,), dtype=torch.float64, device=dev)
stage = (ctypes.c_double * 8)()
for rep in range(2):
    lib.ek_hip_synth_matrix_device(n, 1, dA.data_ptr(), n); lib.ek_hip_synth_matrix_device(n, 2, dB.data_ptr(), n)
    assert lib.ek_hip_solve_device(1, n, n, dA.data_ptr(), n, dB.data_ptr(), n, dw.data_ptr(), dZ.data_ptr(), n, stage, 8) == 0
print("sytrd %.4f" % stage[2])
