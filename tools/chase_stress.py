#!/usr/bin/env python3
"""Stress of the bulge-chasing pipeline's hand-offs (ek_sb2st.hip): d, e and the applied Q2 must come out
bit-identical whatever the timing -- repeated runs, and runs with different numbers of workgroups
(EK_SB2ST_WGS is read per call: 3 workgroups serialise the pipeline almost completely, 256 run it at full
width).  A read of a stale entry, a lost mailbox line or a store that overtakes another shows as a difference.
    chase_stress.py [n ...]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver

lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
dp = ctypes.POINTER(ctypes.c_double)
P = lambda a: a.ctypes.data_as(dp)
B = 64


def random_band(n, seed):
    rng = np.random.default_rng(seed)
    M = np.zeros((n, n))
    for d in range(B + 1):
        v = rng.standard_normal(n - d)
        M[np.arange(d, n), np.arange(0, n - d)] = v
    return M + np.tril(M, -1).T


def sb2st(Bd, ncols):
    n = Bd.shape[0]
    A = np.array(Bd, order="F", copy=True)
    d = np.zeros(n); e = np.zeros(max(n - 1, 1)); flag = ctypes.c_int(-1)
    Z = np.zeros((n, ncols), order="F"); Z[np.arange(ncols) * (n // max(ncols, 1)), np.arange(ncols)] = 1.0
    rc = lib.ek_hip_debug_sb2st(n, P(A), n, P(d), P(e), P(Z), n, ncols, ctypes.byref(flag))
    assert rc == 0 and flag.value == 0, (rc, flag.value)
    return d, e, Z


ok = True
for n in [int(a) for a in sys.argv[1:]] or [130, 777, 1500, 3001, 6000]:
    Bd = random_band(n, n)
    ref = None
    for wgs in (0, 3, 7, 40, 256, 0, 0):
        if wgs: os.environ["EK_SB2ST_WGS"] = str(wgs)
        else: os.environ.pop("EK_SB2ST_WGS", None)
        r = sb2st(Bd, 16)
        if ref is None:
            ref = r
            T = np.diag(r[0]) + np.diag(r[1][:n - 1], 1) + np.diag(r[1][:n - 1], -1)
            ev = np.abs(np.linalg.eigvalsh(Bd) - np.linalg.eigvalsh(T)).max() / np.abs(r[0]).max() if n <= 3001 else 0.0
            print("n=%5d spectrum of T against the band's: %.2e" % (n, ev), flush=True)
            ok &= ev < 8 * n * 2.2e-16
        same = all(np.array_equal(a, b) for a, b in zip(ref, r))
        print("n=%5d workgroups=%-4s identical: %s" % (n, wgs or "auto", same), flush=True)
        ok &= same
    # the position-owned kernel (the default) under shaken timing: pseudo-random pauses of single positions
    os.environ.pop("EK_SB2ST_WGS", None)
    for jit in (0, 1, 7, 12345, 0):
        if jit: os.environ["EK_SB2ST_JITTER"] = str(jit)
        else: os.environ.pop("EK_SB2ST_JITTER", None)
        r = sb2st(Bd, 16)
        same = all(np.array_equal(a, b) for a, b in zip(ref, r))
        print("n=%5d positions in registers, jitter=%-6s identical to the sweep kernel: %s" % (n, jit or "off", same), flush=True)
        ok &= same
    os.environ.pop("EK_SB2ST_JITTER", None)
os.environ.pop("EK_SB2ST_WGS", None)
print("STRESS OK" if ok else "STRESS FAILED", flush=True)
sys.exit(0 if ok else 1)
