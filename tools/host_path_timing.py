#!/usr/bin/env python3
"""Wall time of the host-array entry point ek_hip_solve (PCIe staging included) vs the device
stage times it reports: the PCIe-inclusive rate of DESIGN.md section 5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigenkernel_amd import solver
from oracle import ek_oracle as ok
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
lib = solver.load_library(); lib.ek_hip_init(0)
A = ok.synth_matrix(n, 1); B = ok.synth_matrix(n, 2)
solver.eigen_solver("general_hip", A[:512, :512].copy(), B[:512, :512].copy())
for rep in range(2):
    t = time.time(); ep, _ = solver.eigen_solver("general_hip", A, B); dt = time.time() - t
    dev = sum(v for k, v in ep.stage_seconds.items() if "copies" not in k)
    print("pin=%s N=%d GEP: wall %.3f s (incl. numpy copies), device stages %.3f s, reported copies %.3f s"
          % (os.environ.get("EK_HIP_PIN", "1"), n, dt, dev, ep.stage_seconds["ek_hip:host_device_copies"]))
