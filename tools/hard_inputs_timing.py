#!/usr/bin/env python3
"""Stage seconds of a standard solve for the input classes whose panels CholeskyQR2 cannot factor (they take the
per-panel Householder rescue of ek_sy2sb.hip) beside the dense synthetic matrix: hard_inputs_timing.py [n]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eigenkernel_amd import solver as hip  # noqa: E402
from oracle import ek_oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(1)


def make(kind):
    if kind == "dense":
        return ek_oracle.synth_matrix(n, 1)
    if kind == "banded":
        A = np.zeros((n, n), order="F")
        for d in range(5):
            v = rng.standard_normal(n - d)
            A[np.arange(d, n), np.arange(0, n - d)] = v
            A[np.arange(0, n - d), np.arange(d, n)] = v
        return A
    if kind == "band65":      # one subdiagonal more than the band: every panel a random triangle (condition ~1e7): all rescued
        A = np.zeros((n, n), order="F")
        for d in range(66):
            v = rng.standard_normal(n - d)
            A[np.arange(d, n), np.arange(0, n - d)] = v
            A[np.arange(0, n - d), np.arange(d, n)] = v
        return A
    if kind == "diagonal":
        return np.asfortranarray(np.diag(rng.uniform(1, 2, n)))
    if kind == "low_rank_plus_identity":
        u = rng.standard_normal((n, 3))
        return np.asfortranarray(u @ u.T + np.eye(n))
    if kind == "sparse_pattern":
        A = np.zeros((n, n), order="F")
        i = np.repeat(np.arange(n), 4); j = rng.integers(0, n, 4 * n); v = rng.standard_normal(4 * n)
        A[i, j] = v
        A = np.tril(A); A = np.asfortranarray(A + np.tril(A, -1).T)
        A[np.arange(n), np.arange(n)] = rng.uniform(2, 3, n)
        return A
    raise ValueError(kind)


lib = hip.load_library()
st = (ctypes.c_double * 8)()
base = None
for kind in ("dense", "sparse_pattern", "banded", "band65", "diagonal", "low_rank_plus_identity"):
    A = make(kind)
    for rep in range(2):
        t0 = time.time()
        ep, _ = hip.eigen_solver("hip", A)
        wall = time.time() - t0
    dev = sum(v for k, v in ep.stage_seconds.items() if "copies" not in k)
    lib.ek_hip_debug_last_solve_stats(st, 8)
    if base is None:
        base = dev
    # a cheap acceptance check on a few pairs
    idx = np.linspace(0, n - 1, 8).astype(int)
    r = np.abs(A @ ep.Vectors[:, idx] - ep.Vectors[:, idx] * ep.values[idx]).max()
    print("n=%d %-24s device stages %.3f s (%.2fx dense)  tridiagonalisation %.3f s  two-stage %d  rescued panels %d  max|Av-lv| %.1e"
          % (n, kind, dev, dev / base, ep.stage_seconds["eigen_solver_scalapack_all:pdsytrd"], int(st[1]), int(st[2]), r), flush=True)
