#!/usr/bin/env python3
"""Randomised size sweep of the whole path against the CPU oracle (robustness shake-out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigenkernel_amd import solver
from oracle import ek_oracle as ok
EPS = 2.22e-16
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
sizes = sorted(set([int(x) for x in rng.integers(1, 900, 45)] + [127, 128, 129, 255, 256, 257, 383, 385, 511, 513, 640, 767, 769]))
lib = solver.load_library(); lib.ek_hip_init(0)
bad = 0
for n in sizes:
    A = ok.synth_matrix(n, 1 + n % 5); B = ok.synth_matrix(n, 7)
    for gep in (False, True):
        w_or, _, info, _ = ok.solve(A, B if gep else None)
        nv = int(rng.integers(1, n + 1))
        name = ("general_hip" if gep else "hip") + ("_select" if nv < n else "")
        ep, _ = solver.eigen_solver(name, A, B if gep else None, n_vec=nv if nv < n else None)
        Z = ep.Vectors[:, :nv]; w = ep.values[:nv]
        e1 = np.abs(w - w_or[:nv]).max() / (n * EPS * max(np.abs(w_or).max(), 1e-300))
        R = A @ Z - ((B @ Z) if gep else Z) * w
        e2 = np.abs(R).max() / (n * EPS * np.abs(A).max())
        G = Z.T @ ((B @ Z) if gep else Z) - np.eye(nv)
        e3 = np.abs(G).max() / (n * EPS)
        flag = "" if (e1 <= 8 and e2 <= 64 and e3 <= 64) else "  <-- BAD"
        bad += bool(flag)
        print("n=%4d nv=%4d %-18s dlam %.2f  res %.2f  orth %.2f  (units of n*eps)%s" % (n, nv, name, e1, e2, e3, flag))
print("BAD:", bad)
sys.exit(1 if bad else 0)
