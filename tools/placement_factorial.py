#!/usr/bin/env python3
"""Factorial look at what decides the placement mode (tools): the matrix and the scratch of the
tridiagonalisation in their own allocations of various sizes or inside big blocks."""
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
A_own = alloc(2 * GiB + MiB)
X = alloc(16 * GiB); Y = alloc(16 * GiB)
print("A own 2 GiB | scratch own, size 64 MB .. 8 GiB:", " ".join("%s:%.2f" % (lbl, run(A_own, alloc(sz)))
      for lbl, sz in (("64M", wb + MiB), ("128M", 128 * MiB), ("256M", 256 * MiB), ("512M", 512 * MiB), ("1G", GiB), ("2G", 2 * GiB), ("4G", 4 * GiB), ("8G", 8 * GiB))), flush=True)
print("A own 2 GiB | scratch in block X (0, 7, 15 GiB):", " ".join("%.2f" % run(A_own, X + o * GiB) for o in (0, 7, 15)), flush=True)
print("A in X (0) | scratch in X (4, 8, 15 GiB):       ", " ".join("%.2f" % run(X, X + o * GiB) for o in (4, 8, 15)), flush=True)
print("A in X (0) | scratch in Y (0, 8, 15 GiB):       ", " ".join("%.2f" % run(X, Y + o * GiB) for o in (0, 8, 15)), flush=True)
print("A in X (8 GiB) | scratch in X (0, 4, 15 GiB):   ", " ".join("%.2f" % run(X + 8 * GiB, X + o * GiB) for o in (0, 4, 15)), flush=True)
print("A in X (0) | scratch own 64 MB, own 1 GiB:      ", "%.2f %.2f" % (run(X, alloc(wb + MiB)), run(X, alloc(GiB))), flush=True)
for sz in (3, 4, 6, 8):
    a = alloc(sz * GiB)
    print("A own %d GiB | scratch own 64 MB, in Y:         " % sz, "%.2f %.2f" % (run(a, alloc(wb + MiB)), run(a, Y + 3 * GiB)), flush=True)
