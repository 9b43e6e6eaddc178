#!/usr/bin/env python3
"""GPU check of the two-stage tridiagonalisation pieces (ek_hip_debug_sy2sb / _sb2st) against numpy:
orthogonal similarity, band structure, spectrum; then timings.  Usage: two_stage_check.py [sizes...]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver
from oracle import ek_oracle

lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
dp = ctypes.POINTER(ctypes.c_double)
B = 64
P = lambda a: a.ctypes.data_as(dp)


def sy2sb(A):
    n = A.shape[0]
    A = np.array(A, order="F", copy=True)
    V = np.zeros((n, n), order="F"); tau = np.zeros(n); flag = ctypes.c_int(-1)
    rc = lib.ek_hip_debug_sy2sb(n, P(A), n, P(V), n, P(tau), ctypes.byref(flag))
    assert rc == 0, rc
    return A, V, tau, flag.value


def sb2st(Bd, Z=None):
    n = Bd.shape[0]
    Bd = np.array(Bd, order="F", copy=True)
    d = np.zeros(n); e = np.zeros(max(n - 1, 1)); flag = ctypes.c_int(-1)
    if Z is None:
        rc = lib.ek_hip_debug_sb2st(n, P(Bd), n, P(d), P(e), None, n, 0, ctypes.byref(flag))
    else:
        Z = np.array(Z, order="F", copy=True)
        rc = lib.ek_hip_debug_sb2st(n, P(Bd), n, P(d), P(e), P(Z), n, Z.shape[1], ctypes.byref(flag))
    assert rc == 0, rc
    return d, e[:n - 1], Z, flag.value


def check_sy2sb(n, kind="synth"):
    A = ek_oracle.synth_matrix(n, 1)
    if kind == "wide":
        rng = np.random.default_rng(n); A = rng.standard_normal((n, n)); A = A + A.T
    Ab, V, tau, flag = sy2sb(A)
    L = np.tril(Ab)
    band = np.tril(L) - np.tril(L, -(B + 1))
    Bd = band + np.tril(band, -1).T
    # reflectors: Q1 = H_0 H_1 ...; columns of V
    Q = np.eye(n)
    for j in range(n - 1, -1, -1):
        if tau[j] != 0.0:
            v = V[:, j]
            Q -= tau[j] * np.outer(v, v @ Q)
    anorm = np.linalg.norm(A)
    sim = np.linalg.norm(Q.T @ A @ Q - Bd) / anorm
    orth = np.linalg.norm(Q.T @ Q - np.eye(n))
    w0 = np.linalg.eigvalsh(A); w1 = np.linalg.eigvalsh(Bd)
    ev = np.abs(w0 - w1).max() / np.abs(w0).max()
    # what lies below the band inside A must be R's complement = zero
    below = np.abs(np.tril(Ab, -(B + 1))).max() if n > B + 1 else 0.0
    print("sy2sb n=%5d %-5s flag=%d  |Q^T A Q - Bd|/|A|=%.2e  |Q^TQ-I|=%.2e  eig=%.2e  below-band=%.1e"
          % (n, kind, flag, sim, orth, ev, below), flush=True)
    return Bd, (flag == 0 and sim < 1e-13 and orth < 1e-12 and ev < 1e-13 and below == 0.0)


def check_sb2st(Bd, tag=""):
    n = Bd.shape[0]
    d, e, Q2, flag = sb2st(Bd, np.eye(n))
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    nrm = max(np.linalg.norm(Bd), 1e-300)
    sim = np.linalg.norm(Q2.T @ Bd @ Q2 - T) / nrm
    orth = np.linalg.norm(Q2.T @ Q2 - np.eye(n))
    w0 = np.linalg.eigvalsh(Bd); w1 = np.linalg.eigvalsh(T)
    ev = np.abs(w0 - w1).max() / max(np.abs(w0).max(), 1e-300)
    print("sb2st n=%5d %s flag=%d  |Q2^T Bd Q2 - T|/|Bd|=%.2e  |Q2^TQ2-I|=%.2e  eig=%.2e"
          % (n, tag, flag, sim, orth, ev), flush=True)
    return flag == 0 and sim < 1e-13 and orth < 1e-12 and ev < 1e-13


def random_band(n, seed):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    M = np.tril(M) - np.tril(M, -(B + 1))
    return M + np.tril(M, -1).T


ok = True
sizes = [int(x) for x in sys.argv[1:]] or [3, 5, 64, 65, 66, 100, 129, 130, 200, 321, 400, 700, 1000]
for n in sizes:
    ok &= check_sb2st(random_band(n, n), "band")
for n in sizes:
    if n >= 3:
        Bd, good = check_sy2sb(n)
        ok &= good
        ok &= check_sb2st(Bd, "from-sy2sb")
Bd, good = check_sy2sb(700, "wide"); ok &= good
# bitwise reproducibility of the chase (any stale read would show here)
Bd = random_band(777, 1)
r0 = sb2st(Bd)
for _ in range(3):
    r1 = sb2st(Bd)
    same = np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1], r1[1])
    print("sb2st reproducible:", same, flush=True)
    ok &= same
print("ALL OK" if ok else "FAILURES", flush=True)
sec = (ctypes.c_double * 4)(); flag = ctypes.c_int(0)
for n in (2048, 4096, 8192, 16384):
    if os.environ.get("EK_TS_MAXN") and n > int(os.environ["EK_TS_MAXN"]):
        break
    rc = lib.ek_hip_debug_two_stage_timing(n, n, 1, sec, ctypes.byref(flag))
    rc = lib.ek_hip_debug_two_stage_timing(n, n, 2, sec, ctypes.byref(flag))
    print("timing n=%5d rc=%d flag=%d  sy2sb %.4f s  sb2st %.4f s  q2 %.4f s  q1 %.4f s  total %.4f s"
          % (n, rc, flag.value, sec[0], sec[1], sec[2], sec[3], sum(sec)), flush=True)
sys.exit(0 if ok else 1)
