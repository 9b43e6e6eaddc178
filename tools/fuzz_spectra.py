#!/usr/bin/env python3
"""Spectra and couplings that stress the deflation of the divide & conquer, the splitting of the bulge chasing and the
reduction of the generalized problem: clusters, exact multiplicities, decoupled blocks, ill-conditioned B.  Residual and
orthogonality in units of n eps (the reference's acceptance quantities), eigenvalues against LAPACK.
    python tools/fuzz_spectra.py [n] [seed]"""
import os
import sys

import numpy as np
import scipy.linalg as sl

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from eigenkernel_amd import solver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
EPS = 2.22e-16
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0


def with_spectrum(w):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (Q * w) @ Q.T
    return (A + A.T) / 2


std = []
std.append(("two clusters", with_spectrum(np.concatenate([1 + 1e-13 * rng.standard_normal(n // 2), 2 + 1e-13 * rng.standard_normal(n - n // 2)]))))
std.append(("all equal", with_spectrum(np.full(n, 3.0))))
std.append(("multiplicity 100 + spread", with_spectrum(np.concatenate([np.full(100, -1.0), np.linspace(0, 1, n - 100)]))))
std.append(("geometric 1 .. 1e-14", with_spectrum(np.logspace(0, -14, n))))
std.append(("wilkinson-like pairs", with_spectrum(np.repeat(np.linspace(1, 2, n // 2), 2)[:n] + 1e-15 * rng.standard_normal(n))))
blk = np.zeros((n, n)); h = n // 3
for a, b in ((0, h), (h, 2 * h), (2 * h, n)):
    M = rng.standard_normal((b - a, b - a)); blk[a:b, a:b] = M + M.T
std.append(("three decoupled blocks", blk))
M = rng.standard_normal((n, n)); M = M + M.T; M[n // 2:, :n // 2] *= 1e-18; M[:n // 2, n // 2:] *= 1e-18
std.append(("weakly coupled halves", M))
Z0 = np.zeros((n, n)); Z0[: n // 4, : n // 4] = (lambda G: G + G.T)(rng.standard_normal((n // 4, n // 4)))
std.append(("three quarters zero", Z0))
T = np.diag(rng.standard_normal(n)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
std.append(("tridiagonal", T))
std.append(("1e150 scaled", 1e150 * (lambda G: G + G.T)(rng.standard_normal((n, n)))))
std.append(("1e-150 scaled", 1e-150 * (lambda G: G + G.T)(rng.standard_normal((n, n)))))
bad = 0
for name, A in std:
    A = np.asfortranarray(A)
    w0 = np.linalg.eigvalsh(A)
    ep, _ = solver.eigen_solver("hip", A)
    Z = ep.Vectors; sc = max(np.abs(w0).max(), 1e-300)
    err = np.abs(ep.values - w0).max() / (n * EPS * sc)
    res = np.abs(A @ Z - Z * ep.values).max() / (n * EPS * sc)
    orth = np.abs(Z.T @ Z - np.eye(n)).max() / (n * EPS)
    flag = "" if (err <= 4 and res <= 16 and orth <= 16) else "   <-- BAD"
    bad += bool(flag)
    print("SEP %-28s dlam %6.2f  res %6.2f  orth %6.2f  (n eps)%s" % (name, err, res, orth, flag), flush=True)
# the *_select arms where the cut falls inside a cluster, at a multiplicity, at the ends
for name, A in std[:5]:
    A = np.asfortranarray(A); w0 = np.linalg.eigvalsh(A); sc = max(np.abs(w0).max(), 1e-300)
    for nv in (1, 37, n // 2, n - 1):
        ep, _ = solver.eigen_solver("hip_select", A, n_vec=nv)
        Z = ep.Vectors[:, :nv]; w = ep.values[:nv]
        err = np.abs(w - w0[:nv]).max() / (n * EPS * sc)
        res = np.abs(A @ Z - Z * w).max() / (n * EPS * sc)
        orth = np.abs(Z.T @ Z - np.eye(nv)).max() / (n * EPS)
        flag = "" if (err <= 4 and res <= 16 and orth <= 16) else "   <-- BAD"
        bad += bool(flag)
        if flag: print("SEL %-28s n_vec %5d dlam %6.2f  res %6.2f  orth %6.2f  (n eps)%s" % (name, nv, err, res, orth, flag), flush=True)
print("select arms on the first five spectra done", flush=True)
A = (lambda G: G + G.T)(rng.standard_normal((n, n)))
for kb in (1e2, 1e5, 1e8, 1e11):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    B = (Q * np.logspace(0, -np.log10(kb), n)) @ Q.T; B = (B + B.T) / 2
    w0, Z0 = sl.eigh(A, B)                                   # LAPACK on the same pencil: the yardstick
    ep, _ = solver.eigen_solver("general_hip", np.asfortranarray(A), np.asfortranarray(B))

    def quantities(w, Z):
        R = A @ Z - (B @ Z) * w
        res = (np.abs(R).max(axis=0) / (np.abs(A).max() + np.abs(w) * np.abs(B).max())).max() / (n * EPS)
        return res, np.abs(Z.T @ B @ Z - np.eye(n)).max() / (n * EPS)
    res, orth = quantities(ep.values, ep.Vectors)
    res0, orth0 = quantities(w0, Z0)
    # eigenvalues of a pencil with cond(B) = kb are defined to eps kb |lambda| at best
    err = (np.abs(ep.values - w0) / np.maximum(np.abs(w0), 1.0)).max() / (n * EPS * kb)
    flag = "" if (res <= 4 * max(res0, 16) and orth <= 4 * max(orth0, 16) and err <= 4) else "   <-- BAD"
    bad += bool(flag)
    print("GEP cond(B) %.0e  dlam/(n eps cond |lam|) %8.2e  res %10.2f (LAPACK %10.2f)  orth %10.2f (LAPACK %10.2f)  (n eps)%s" %
          (kb, err, res, res0, orth, orth0, flag), flush=True)
# the reference's family: sparse / banded Hamiltonian A with a banded, diagonally dominant overlap B
def banded(hw, diag):
    M = np.tril(rng.standard_normal((n, n))); M = M - np.tril(M, -(hw + 1)); M = M + np.tril(M, -1).T
    return M + diag * np.eye(n)


pairs = [("A band 5, B band 5", banded(5, 0.0), 0.05 * banded(5, 0.0) + np.eye(n)),
         ("A band 40, B = I", banded(40, 0.0), np.eye(n)),
         ("A band 3, B diagonal", banded(3, 0.0), np.diag(rng.uniform(0.5, 2.0, n))),
         ("A = B", None, None), ("A = 0", np.zeros((n, n)), 0.1 * banded(8, 0.0) + 2 * np.eye(n)),
         ("A dense, B band 64 decaying", (lambda G: G + G.T)(rng.standard_normal((n, n))),
          np.exp(-np.abs(np.subtract.outer(np.arange(n), np.arange(n))) / 3.0) * (np.abs(np.subtract.outer(np.arange(n), np.arange(n))) <= 64))]
for name, A2, B2 in pairs:
    if A2 is None:
        B2 = 0.1 * banded(8, 0.0) + 2 * np.eye(n); A2 = B2.copy()
    w0, Z0 = sl.eigh(A2, B2)
    ep, _ = solver.eigen_solver("general_hip", np.asfortranarray(A2), np.asfortranarray(B2))
    sc = max(np.abs(w0).max(), 1e-300)
    err = np.abs(ep.values - w0).max() / (n * EPS * max(sc, 1.0))
    Z = ep.Vectors
    R = A2 @ Z - (B2 @ Z) * ep.values
    res = np.abs(R).max() / (n * EPS * max(np.abs(A2).max(), np.abs(B2).max() * sc, 1e-300))
    orth = np.abs(Z.T @ B2 @ Z - np.eye(n)).max() / (n * EPS)
    flag = "" if (err <= 8 and res <= 16 and orth <= 16) else "   <-- BAD"
    bad += bool(flag)
    print("GEP %-30s dlam %6.2f  res %6.2f  orth %6.2f  (n eps)%s" % (name, err, res, orth, flag), flush=True)
print("BAD:", bad)
sys.exit(1 if bad else 0)
