#!/usr/bin/env python3
"""Which array's position decides the placement mode of the tridiagonalisation (tools)?
Blocks of 21 GiB (the library's arena of ~13 GiB + 8 GiB of slack); the matrix (2 GiB at N=16384),
the stage scratch and the vectors are put at chosen offsets; time of the first 64 columns."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver
n = 16384
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
GiB, MiB = 1 << 30, 1 << 20
wb = int(lib.ek_hip_debug_sytrd_work_bytes(n))
arena = 13 * GiB + 64 * MiB
work_off = 8 * GiB + 32 * MiB
slack = 8 * GiB
total = arena + slack
sec = ctypes.c_double(0)
hold = []
for b in range(3):
    blk = ctypes.c_void_p()
    assert lib.ek_hip_malloc(ctypes.byref(blk), total) == 0
    hold.append(blk)
    base = blk.value
    def run(offA, offW, offV):
        assert lib.ek_hip_debug_sytrd_at(n, 64, 3, ctypes.c_void_p(base + offA), ctypes.c_void_p(base + offW),
                                         ctypes.c_void_p(base + offV), ctypes.byref(sec)) == 0
        return sec.value * 1e3
    rows = [("library layout, shift 0", 0, work_off, arena - MiB),
            ("library layout, shift 8 GiB (top)", slack, slack + work_off, slack + arena - MiB),
            ("A as at top, work+vecs as at shift 0", slack, work_off, arena - MiB),
            ("A as at shift 0, work+vecs as at top", 0, slack + work_off, slack + arena - MiB),
            ("A as at shift 0, work at shift 0, vecs at top", 0, work_off, slack + arena - MiB),
            ("A as at shift 0, work at top, vecs at shift 0", 0, slack + work_off, arena - MiB)]
    for name, a, w, v in rows:
        print("block %d: %-48s %.3f ms" % (b, name, run(a, w, v)), flush=True)
