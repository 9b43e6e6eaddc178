#!/usr/bin/env python3
"""Which part of the tridiagonalisation's scratch decides the placement mode (tools)?  Finds a
same-colour (slow) and a different-colour (fast) scratch for one matrix, then moves sub-buffers one by
one from the slow scratch into the fast one (mask: 1 x, 2 panel, 4 row-part sums, 8 column-part sums,
16 the rest) and times the first 64 columns."""
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
V = alloc(MiB); A = alloc(2 * GiB + MiB)
def run(w):
    assert lib.ek_hip_debug_sytrd_at(n, 64, 3, ctypes.c_void_p(A), ctypes.c_void_p(w), ctypes.c_void_p(V), ctypes.byref(sec)) == 0
    return sec.value * 1e3
slow = fast = None
for _ in range(12):
    w = alloc(wb + MiB); t = run(w)
    if t > 13.3 and slow is None: slow = w
    if t < 13.15 and fast is None: fast = w
    if slow and fast: break
assert slow and fast, "both colours not found"
print("slow scratch %.3f ms, fast scratch %.3f ms" % (run(slow), run(fast)))
for mask, name in ((1, "x"), (2, "panel"), (4, "row-part sums"), (8, "column-part sums"), (16, "rest"),
                   (12, "both partial sums"), (3, "x + panel"), (31, "all")):
    lib.ek_hip_debug_sytrd_split(ctypes.c_void_p(fast), mask)
    a = run(slow)
    lib.ek_hip_debug_sytrd_split(ctypes.c_void_p(slow), mask)
    b = run(fast)
    print("%-20s moved slow->fast: %.3f ms   moved fast->slow: %.3f ms" % (name, a, b), flush=True)
lib.ek_hip_debug_sytrd_split(None, 0)
