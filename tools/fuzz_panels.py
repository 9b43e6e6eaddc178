#!/usr/bin/env python3
"""Standard problems whose dense -> band panels are moderately ill-conditioned (the range between what CholeskyQR2 takes
and what its device-side check sends to the Householder rescue): eigenvalues against numpy in units of n eps max|lambda|.
    python tools/fuzz_panels.py [n] [seed]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eigenkernel_amd import solver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
EPS = 2.22e-16
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
st = (ctypes.c_double * 8)()


def sym(M):
    return np.tril(M) + np.tril(M, -1).T


cases = []
for hw in (65, 70, 96, 128, 200, 400):
    M = np.tril(rng.standard_normal((n, n))); cases.append(("band %d" % hw, sym(M - np.tril(M, -(hw + 1)))))
for k in (2, 4, 6, 8, 10):
    D = 10.0 ** (-k * np.arange(n) / n); G = rng.standard_normal((n, n)); cases.append(("graded 1e-%d" % k, sym(D[:, None] * G * D[None, :])))
for noise in (1e-3, 1e-5, 1e-7, 1e-9, 1e-11):
    A = sym(rng.standard_normal((n, n)))
    for j in range(1, 40, 2):                     # pairs of nearly parallel columns below the first band
        A[64:, j] = A[64:, j - 1] + noise * rng.standard_normal(n - 64); A[j, 64:] = A[64:, j]
    cases.append(("parallel columns %.0e" % noise, A))
for p in (0.5, 0.1, 0.02):
    Mk = rng.random((n, n)) < p; A = sym(rng.standard_normal((n, n)) * Mk); cases.append(("random pattern %.2f" % p, A))
x = np.linspace(0, 1, n); cases.append(("hilbert-like kernel", 1.0 / (1.0 + np.abs(x[:, None] - x[None, :]) * n) ** 0.5))
cases.append(("exp kernel", np.exp(-np.abs(x[:, None] - x[None, :]) * 30)))
cases.append(("rank 70 + 1e-8 noise", (lambda U: U @ U.T + 1e-8 * sym(rng.standard_normal((n, n))))(rng.standard_normal((n, 70)))))
bad = 0
for name, A in cases:
    A = np.asfortranarray(A)
    w0 = np.linalg.eigvalsh(A)
    ep, _ = solver.eigen_solver("hip", A)
    lib.ek_hip_debug_last_solve_stats(st, 8)
    err = np.abs(ep.values - w0).max() / (n * EPS * max(np.abs(w0).max(), 1e-300))
    Z = ep.Vectors
    res = np.abs(A @ Z - Z * ep.values).max() / (n * EPS * max(np.abs(w0).max(), 1e-300))
    orth = np.abs(Z.T @ Z - np.eye(n)).max() / (n * EPS)
    flag = "" if (err <= 4 and res <= 16 and orth <= 16) else "   <-- BAD"
    bad += bool(flag)
    print("%-28s dlam %6.2f  res %6.2f  orth %6.2f  (n eps)   two-stage %d rescued %3d band %d%s" %
          (name, err, res, orth, int(st[1]), int(st[2]), int(st[3]), flag), flush=True)
print("BAD:", bad)
sys.exit(1 if bad else 0)
