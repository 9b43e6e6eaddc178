#!/usr/bin/env python3
"""fp64 TFLOP/s of the library's GEMM (ek_gemm.hip) on the operand shapes the path issues at a given order
(default 16384), against the datasheet peak (78.6) and the ceiling tools/mfma_peak measures (49.5):
    python tools/gemm_shapes.py [n] > profiles/r02_gemm_shapes_n<n>.txt"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
GiB = 1 << 30
sec = ctypes.c_double(0)
def alloc(nbytes):
    p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), int(nbytes)) == 0; return p
A, B, C = (alloc(n * n * 8 + (1 << 20)) for _ in range(3))
for b in (A, B, C):
    assert lib.ek_hip_synth_matrix_device(n, 1, b, n) == 0
h = n // 2
shapes = [
    ("SYR2K of a panel, one stage (K=128, lower)          C(n,n) -= P1 P2^T", 0, 1, n, n, 128, 1.0, 1, 0.5),
    ("SYR2K of the dense->band stage at m = n/2           C(m,m) -= [W|V][V|W]^T", 0, 1, h, h, 128, 1.0, 1, 0.5),
    ("the same update for TWO panels at once (K = 256)     C(n,n) -= [W1 V1 W2 V2][V1 W1 V2 W2]^T", 0, 1, n, n, 256, 1.0, 1, 0.5),
    ("... K = 512                                          (four panels)", 0, 1, n, n, 512, 1.0, 1, 0.5),
    ("... K = 128 without reading C (beta = 0)             what the read of C costs", 0, 1, n, n, 128, 0.0, 1, 0.5),
    ("... K = 64                                           the floor of a pass over the triangle", 0, 1, n, n, 64, 1.0, 1, 0.5),
    ("back-transformation, W1 = V_b^T Z                   (512 x n) = (n x 512)^T (n x n)", 1, 0, 512, n, n, 0.0, 0, 1.0),
    ("back-transformation, Z -= V_b W2                     (n x n) -= (n x 512)(512 x n)", 0, 0, n, n, 512, 1.0, 0, 1.0),
    ("triangular solve / Cholesky update, half size        (n/2)^3, C -= A B", 0, 0, h, h, h, 1.0, 0, 1.0),
    ("triangular solve, wide right-hand side               (n/2 x n) -= (n/2 x n/2)(n/2 x n)", 0, 0, h, n, h, 1.0, 0, 1.0),
    ("SYRK of the Cholesky factorisation, half size        C(n/2,n/2) -= A A^T (lower)", 0, 1, h, h, h, 1.0, 1, 0.5),
    ("D&C top merge                                        (n/2 x n) = (n/2 x n/2)(n/2 x n), beta = 0", 0, 0, h, n, h, 0.0, 0, 1.0),
    ("Gram matrices of the back-transformation             (512 x 512) = (n x 512)^T (n x 512)", 1, 0, 512, 512, n, 0.0, 0, 1.0),
    ("small-K panel product                                (n x 64) = (n x n/4 slice)...(K = 2048)", 0, 0, n, 64, 2048, 0.0, 0, 1.0),
]
print("GEMM shapes of the path at n = %d (3 repetitions each, HIP events); fraction of the 78.6 TFLOP/s datasheet peak" % n)
print("%-95s %10s %9s %7s" % ("shape", "ms", "TFLOP/s", "frac"))
for name, ta, tb, m, nn, k, beta, lower, share in shapes:
    rc = lib.ek_hip_debug_gemm_at(ta, tb, m, nn, k, A, n, B, n, beta, C, n, lower, 3, ctypes.byref(sec))
    assert rc == 0, rc
    fl = 2.0 * m * nn * k * share          # lower-only updates do half the tiles (plus the diagonal ones)
    print("%-95s %10.3f %9.1f %7.2f" % (name, sec.value * 1e3, fl / sec.value / 1e12, fl / sec.value / 1e12 / 78.6), flush=True)
