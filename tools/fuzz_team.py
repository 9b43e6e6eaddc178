#!/usr/bin/env python3
"""Randomised sweep of the distributed stages in team-rehearsal form (tools, not product).

  python tools/fuzz_team.py [cases] [seed]

Random orders (2..1600, biased to strip boundaries) and team sizes (1..16): PDSYTRD, PDPOTRF,
PDSYGST and the dense -> band stage on a 1 x P grid against the single-GPU stages; every rank of a team must end with the same
bits.  Prints one line per failure and a summary.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver as hip   # noqa: E402
from oracle import ek_oracle                 # noqa: E402

EPS = 2.220446049250313e-16
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
lib = hip.load_library()
assert lib.ek_hip_init(0) == 0
bad = 0
for c in range(cases):
    if rng.random() < 0.5:
        n = int(128 * rng.integers(1, 12) + rng.integers(-2, 3))
    else:
        n = int(rng.integers(2, 1600))
    n = max(2, n)
    P = int(rng.integers(1, 17))
    A = ek_oracle.synth_matrix(n, 1)
    B = ek_oracle.synth_matrix(n, 2)
    msgs = []
    Ar, d, e, tau, info, mm = hip.sytrd_team(A, P)
    Ar1, d1, e1, tau1, info1 = hip.sytrd(A)
    if info or info1 or mm:
        msgs.append("sytrd info %d/%d mismatch %d" % (info, info1, mm))
    else:
        T = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
        T1 = np.diag(d1) + np.diag(e1, -1) + np.diag(e1, 1)
        err = np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(T1)).max()
        if not err <= 8 * n * EPS * np.abs(A).max() * 4:
            msgs.append("sytrd spectrum %.2e" % err)
    L, info, mm = hip.potrf_team(B, P)
    L1, info1 = hip.potrf(B)
    if info or info1 or mm:
        msgs.append("potrf info %d/%d mismatch %d" % (info, info1, mm))
    elif not np.abs(np.tril(L) - np.tril(L1)).max() <= 16 * n * EPS * np.abs(L1).max():
        msgs.append("potrf diff %.2e" % np.abs(np.tril(L) - np.tril(L1)).max())
    Lt = np.tril(L1)
    C, info = hip.sygst_team(A, Lt, P)
    C1, info1 = hip.sygst(A, Lt)
    il = np.tril_indices(n)
    if info or info1:
        msgs.append("sygst info %d/%d" % (info, info1))
    elif not np.abs(C[il] - C1[il]).max() <= 64 * n * EPS * np.abs(C1[il]).max():
        msgs.append("sygst diff %.2e" % np.abs(C[il] - C1[il]).max())
    if n >= 3:     # the team form of the dense -> band stage: members identical, spectrum of the band = spectrum of A
        Ab, V, tau1s, flag, mm = hip.sy2sb_team(A, P)
        if flag & 0xff or mm:
            msgs.append("sy2sb_team flag %d mismatch %d" % (flag, mm))
        else:
            Lb = np.tril(Ab) - np.tril(Ab, -65)
            Bd = Lb + np.tril(Lb, -1).T
            w0 = np.linalg.eigvalsh(A)
            err = np.abs(w0 - np.linalg.eigvalsh(Bd)).max()
            if not err <= 8 * n * EPS * np.abs(w0).max() or np.abs(np.tril(Ab, -65)).max() != 0.0:
                msgs.append("sy2sb_team spectrum %.2e" % err)
    if msgs:
        bad += 1
        print("FAIL n=%d P=%d: %s" % (n, P, "; ".join(msgs)), flush=True)
    elif c % 10 == 0:
        print("ok so far: case %d (n=%d P=%d)" % (c, n, P), flush=True)
print("%d bad of %d" % (bad, cases))
