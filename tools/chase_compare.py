#!/usr/bin/env python3
"""The two bulge-chasing kernels of ek_sb2st.hip (sweeps through memory / positions in registers) run the same
arithmetic: d, e and the applied Q2 must agree bit for bit.  chase_compare.py [n ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eigenkernel_amd import solver as hip  # noqa: E402

B = 64


def random_band(n, seed):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    M = np.tril(M) - np.tril(M, -(B + 1))
    return M + np.tril(M, -1).T


def run(Bd, Z0, mode):
    os.environ["EK_SB2ST_CHASE"] = str(mode)
    t0 = time.time()
    out = hip.sb2st(Bd, Z0)
    return out, time.time() - t0


def main():
    ns = [int(a) for a in sys.argv[1:]] or [3, 4, 5, 64, 65, 66, 67, 100, 129, 130, 131, 200, 257, 321, 640, 777, 1000, 1500, 2500]
    bad = 0
    for n in ns:
        Bd = random_band(n, n)
        Z0 = np.eye(n)[:, ::max(n // 16, 1)][:, :16].copy()
        (d1, e1, Z1, f1), t1 = run(Bd, Z0, 1)
        (d2, e2, Z2, f2), t2 = run(Bd, Z0, 2)
        same = np.array_equal(d1, d2) and np.array_equal(e1, e2) and np.array_equal(Z1, Z2)
        T = np.diag(d2) + np.diag(e2, 1) + np.diag(e2, -1)
        err = np.abs(np.linalg.eigvalsh(Bd) - np.linalg.eigvalsh(T)).max() if n <= 3000 else -1.0
        print("n=%5d flags %d %d  identical %s  spectrum err %.2e  (%.3f s, %.3f s)  max|d1-d2| %.2e" %
              (n, f1, f2, same, err, t1, t2, np.abs(d1 - d2).max()), flush=True)
        bad += (not same) or f1 != 0 or f2 != 0
    os.environ.pop("EK_SB2ST_CHASE", None)
    print("FAILED" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
