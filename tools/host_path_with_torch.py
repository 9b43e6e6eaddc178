#!/usr/bin/env python3
"""The host path (ek_hip_solve, PCIe included) in a process that has loaded and initialised PyTorch first -- PyTorch ships
its own HIP runtime (torch/lib/libamdhip64.so), which then serves the library's calls too -- against tools/host_path_trace.py
(the library alone, the system's runtime: what a host of the reference's shape links).
    EK_HIP_PIPE_TRACE=1 python tools/host_path_with_torch.py [n]"""
import os
import sys

import torch
x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from eigenkernel_amd import solver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
print("rep 0: %s" % (bench.host_path_step(lib, solver, 1, n, n),), flush=True)
with open("/proc/self/maps") as f:
    print(sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libhsa-runtime" in l}))
