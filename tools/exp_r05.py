import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
n = 16384
sec = ctypes.c_double(0)
def alloc(nbytes):
    p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), int(nbytes)) == 0; return p
A, B, C = (alloc(n * n * 8 + (1 << 20)) for _ in range(3))
for b in (A, B, C):
    assert lib.ek_hip_synth_matrix_device(n, 1, b, n) == 0
for name, ta, tb, m, nn, k, beta, lower, share in [
    ("K=128 lower beta=1", 0, 1, n, n, 128, 1.0, 1, 0.5),
    ("K=128 lower beta=0", 0, 1, n, n, 128, 0.0, 1, 0.5),
    ("K=128 full  beta=1", 0, 1, n, n, 128, 1.0, 0, 1.0),
    ("K=128 full  beta=0", 0, 1, n, n, 128, 0.0, 0, 1.0),
    ("K=256 lower beta=1", 0, 1, n, n, 256, 1.0, 1, 0.5),
    ("K=256 lower beta=0", 0, 1, n, n, 256, 0.0, 1, 0.5),
    ("K=512 lower beta=1", 0, 1, n, n, 512, 1.0, 1, 0.5),
    ("K=64  lower beta=1", 0, 1, n, n, 64, 1.0, 1, 0.5),
    ("K=128 lower beta=1 m=8192", 0, 1, 8192, 8192, 128, 1.0, 1, 0.5),
    ("K=128 lower beta=1 m=12288", 0, 1, 12288, 12288, 128, 1.0, 1, 0.5),
]:
    rc = lib.ek_hip_debug_gemm_at(ta, tb, m, nn, k, A, n, B, n, beta, C, n, lower, 5, ctypes.byref(sec))
    assert rc == 0, rc
    fl = 2.0 * m * nn * k * share
    gb = m * nn * share * 8 * (2 if beta else 1) / 1e9
    print("%-30s %8.3f ms %6.1f TF  C-traffic %.2f TB/s" % (name, sec.value * 1e3, fl / sec.value / 1e12, gb / sec.value / 1e3), flush=True)
