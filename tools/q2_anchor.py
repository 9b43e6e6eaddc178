#!/usr/bin/env python3
"""Digest of Z <- Q2 Z on fixed inputs (to compare two library builds bit for bit): q2_anchor.py [EK_HIP_LIB set outside].
tests/golden/q2_anchor_digests.txt holds its output (round 2's pair kernel and every form since give these bits)."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eigenkernel_amd import solver  # noqa: E402

lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
B = 64
for n, ncols in [(3, 3), (66, 66), (130, 17), (321, 64), (1000, 1000), (1500, 333), (2500, 700)]:
    M = np.random.RandomState(7 * n + 3).standard_normal((n, n))      # (RandomState: a frozen stream, unlike Generator's)
    M = np.tril(M) - np.tril(M, -(B + 1))
    Bd = M + np.tril(M, -1).T
    Z0 = np.random.RandomState(n).standard_normal((n, ncols))
    d, e, Z, f = solver.sb2st(Bd, Z0)
    print(n, ncols, f, hashlib.sha256(np.ascontiguousarray(Z).tobytes()).hexdigest()[:16], flush=True)
