#!/usr/bin/env python3
"""Placement of the tridiagonalisation's SCRATCH (x, panel, partial sums: 64 MB at N=16384) with the
matrix fixed: time of the first 64 columns for scratch positions spread over a 16 GiB block, and for
separately allocated scratch buffers (tools)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver
n = 16384
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
GiB, MiB = 1 << 30, 1 << 20
wb = int(lib.ek_hip_debug_sytrd_work_bytes(n))
sec = ctypes.c_double(0)
def alloc(nbytes):
    p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), nbytes) == 0; return p
A = alloc(2 * GiB + MiB); V = alloc(MiB)
def run(work_addr):
    assert lib.ek_hip_debug_sytrd_at(n, 64, 3, A, ctypes.c_void_p(work_addr), V, ctypes.byref(sec)) == 0
    return sec.value * 1e3
blk = alloc(16 * GiB)
row = []
for k in range(0, 64):
    off = k * 256 * MiB
    if off + wb > 16 * GiB: break
    row.append(run(blk.value + off))
print("scratch at 256 MiB steps inside one 16 GiB block:", " ".join("%.2f" % t for t in row), flush=True)
row = []
keep = []
for k in range(24):
    p = alloc(wb + MiB); keep.append(p)
    row.append(run(p.value))
print("24 separately allocated scratch buffers:          ", " ".join("%.2f" % t for t in row), flush=True)
row = []
for k in range(16):
    row.append(run(blk.value + 16 * GiB - wb - (k * 4 + 1) * MiB))
print("scratch 1, 5, 9, ... MiB below the top of the block:", " ".join("%.2f" % t for t in row), flush=True)
