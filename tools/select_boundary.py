#!/usr/bin/env python3
"""The `*_select` arms around the boundary where the divide & conquer switches to its compact bases (at most half of the
columns wanted): eigenpairs must be the bits of the full solve's first n_vec columns.
    python tools/select_boundary.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eigenkernel_amd import solver  # noqa: E402
from oracle import ek_oracle as ok  # noqa: E402  (tools may use the oracle's matrix generator)

lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
bad = 0
for n in (1500, 1501, 640, 2049):
    A = ok.synth_matrix(n, 1)
    B = ok.synth_matrix(n, 2)
    for gep in (False, True):
        full, _ = solver.eigen_solver("general_hip" if gep else "hip", A, B if gep else None)
        for nv in sorted({1, 2, n // 2 - 1, n // 2, n // 2 + 1, (n + 1) // 2, (n + 1) // 2 + 1, n - 1}):
            ep, _ = solver.eigen_solver("general_hip_select" if gep else "hip_select", A, B if gep else None, n_vec=nv)
            same = np.array_equal(ep.values[:nv], full.values[:nv]) and np.array_equal(ep.Vectors[:, :nv], full.Vectors[:, :nv])
            if not same:
                dv = np.abs(ep.Vectors[:, :nv] - full.Vectors[:, :nv]).max()
                dl = np.abs(ep.values[:nv] - full.values[:nv]).max()
                bad += 1
                print("n=%d gep=%d n_vec=%d DIFFERENT: dlam %.2e dZ %.2e" % (n, gep, nv, dl, dv), flush=True)
        print("n=%d gep=%d done" % (n, gep), flush=True)
print("BAD:", bad)
