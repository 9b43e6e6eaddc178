#!/usr/bin/env python3
"""Addresses vs placement mode (tools): prints the device addresses of the matrix and scratch and the
time of the first 64 columns for many pairs of allocations."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver
n = 16384
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
GiB, MiB = 1 << 30, 1 << 20
wb = int(lib.ek_hip_debug_sytrd_work_bytes(n))
sec = ctypes.c_double(0)
def alloc(nbytes):
    p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), int(nbytes)) == 0; return p.value
V = alloc(MiB)
def run(a, w):
    assert lib.ek_hip_debug_sytrd_at(n, 64, 3, ctypes.c_void_p(a), ctypes.c_void_p(w), ctypes.c_void_p(V), ctypes.byref(sec)) == 0
    return sec.value * 1e3
As = [("A%d" % i, alloc(sz * GiB + MiB)) for i, sz in enumerate((2, 2, 16, 2, 3, 16, 2))]
Ws = [("W%d" % i, alloc(sz)) for i, sz in enumerate((wb + MiB, wb + MiB, GiB, wb + MiB, 16 * GiB, wb + MiB))]
print("addresses:", " ".join("%s=%#x" % (k, v) for k, v in As + Ws))
print("%-22s" % "A \\ scratch", " ".join("%-8s" % k for k, _ in Ws), " | scratch inside other A blocks: A2+4G A5+4G")
for ka, a in As:
    row = ["%-8.2f" % run(a, w) for _, w in Ws]
    extra = ["%-8.2f" % run(a, As[2][1] + 4 * GiB), "%-8.2f" % run(a, As[5][1] + 4 * GiB)]
    print("%-22s" % ("%s=%#x" % (ka, a)), " ".join(row), " | ", " ".join(extra), flush=True)
