#!/usr/bin/env python3
"""sha256 digests of what the team form of the dense -> band stage (ek_hip_debug_sy2sb_team: the whole team rehearsed on
one GPU) and the single-GPU stage produce, for a few orders and team sizes: the anchor that a restructuring of the
team form (streams, look-ahead) is held to bit for bit.
    python tools/team_digests.py [out.txt]          (EK_HIP_LIB=<other build> for the reference side)"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eigenkernel_amd import solver  # noqa: E402
from oracle import ek_oracle  # noqa: E402  (the synthetic generator only)

CASES = [(700, 2), (700, 3), (1500, 2), (1500, 8), (2600, 4), (2600, 8), (5300, 8), (6500, 3)]


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
    lines = []
    for n, P in CASES:
        A = ek_oracle.synth_matrix(n, 1 + n % 3)
        Ab, V, tau, flag, mism = solver.sy2sb_team(A, P)
        lines.append("team n=%d P=%d flag=%d mismatch=%d %s" % (n, P, flag, mism, digest(np.tril(Ab) - np.tril(Ab, -65), V, tau)))
        print(lines[-1], flush=True)
    for n in sorted(set(c[0] for c in CASES)):
        A = ek_oracle.synth_matrix(n, 1 + n % 3)
        Ab, V, tau, flag = solver.sy2sb(A)
        lines.append("single n=%d flag=%d %s" % (n, flag, digest(np.tril(Ab) - np.tril(Ab, -65), V, tau)))
        print(lines[-1], flush=True)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
