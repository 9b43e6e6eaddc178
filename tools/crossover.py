"""One-stage against two-stage tridiagonalisation through the whole-path call, standard problem, a few orders:
the measurement behind the default of EK_HIP_TWO_STAGE_MIN (ek_comm.hip two_stage_min)."""
import ctypes, sys, time, numpy as np
sys.path.insert(0, '.')
from eigenkernel_amd import solver
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
dp = ctypes.POINTER(ctypes.c_double)
def run(n, gep, force):
    nn = n*n*8
    ptr = lambda: ctypes.c_void_p()
    bufs = []
    def alloc(b):
        p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), b) == 0; bufs.append(p); return p
    dA, dZ, dw = alloc(nn), alloc(nn), alloc(n*8)
    dB = alloc(nn) if gep else None
    lib.ek_hip_debug_set_two_stage(force)
    best = 1e9
    for it in range(3):
        lib.ek_hip_synth_matrix_device(n, 1, dA, n)
        if gep: lib.ek_hip_synth_matrix_device(n, 2, dB, n)
        st = np.zeros(8)
        t0 = time.perf_counter()
        info = lib.ek_hip_solve_device(1 if gep else 0, n, n, dA, n, dB, n, dw, dZ, n, st.ctypes.data_as(dp), 8)
        t1 = time.perf_counter()
        assert info == 0
        best = min(best, t1 - t0)
    for p in bufs: lib.ek_hip_free(p)
    lib.ek_hip_finalize()
    return best, st
import sys as _s
for n in ([int(a) for a in _s.argv[1:]] or [2048, 3072, 4096, 6144, 8192]):
    for gep in (False,):
        a, sa = run(n, gep, 0)
        b, sb = run(n, gep, 1)
        print("n=%5d gep=%d one-stage %.4f s  two-stage %.4f s   sytrd %.4f/%.4f ormtr %.4f/%.4f" % (n, gep, a, b, sa[2], sb[2], sa[5], sb[5]), flush=True)
