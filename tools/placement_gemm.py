#!/usr/bin/env python3
"""Do the MFMA-bound stages care where their operands lie relative to each other (tools)?
Shapes of the path at N=16384 with A, B, C in separate allocations; prints time per call for several
allocation triples (a bimodal spread would mean yes)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
GiB, MiB = 1 << 30, 1 << 20
n = 16384
sec = ctypes.c_double(0)
def alloc(nbytes):
    p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), int(nbytes)) == 0; return p.value
bufs = [alloc(2 * GiB + MiB) for _ in range(6)]
for b in bufs:
    assert lib.ek_hip_synth_matrix_device(n, 1, ctypes.c_void_p(b), n) == 0
def run(ta, tb, m, nn, k, a, b, c, beta, lower, reps=3):
    assert lib.ek_hip_debug_gemm_at(ta, tb, m, nn, k, ctypes.c_void_p(a), n, ctypes.c_void_p(b), n, beta, ctypes.c_void_p(c), n,
                                    lower, reps, ctypes.byref(sec)) == 0
    return sec.value * 1e3
shapes = [("back-transform block  C(16384x16384) -= V(16384x512) T", 0, 0, n, n, 512, 1.0, 0),
          ("rank-128 trailing update, lower (SYR2K)", 0, 1, n, n, 128, 1.0, 1),
          ("half-size product (8192^3), beta=0", 0, 0, 8192, 8192, 8192, 0.0, 0),
          ("solve update C(8192x16384) -= L21 X", 0, 0, 8192, n, 8192, 1.0, 0)]
for name, ta, tb, m, nn, k, beta, lower in shapes:
    row = []
    for (ia, ib, ic) in ((0, 1, 2), (0, 1, 3), (0, 1, 4), (0, 1, 5), (2, 3, 0), (2, 3, 1), (4, 5, 0), (0, 0, 1), (0, 1, 1)):
        row.append("%.3f" % run(ta, tb, m, nn, k, bufs[ia], bufs[ib], bufs[ic], beta, lower))
    print("%-60s %s ms" % (name, " ".join(row)), flush=True)
