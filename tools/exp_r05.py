import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
n = 16384
def alloc(nbytes):
    p = ctypes.c_void_p(); assert lib.ek_hip_malloc(ctypes.byref(p), int(nbytes)) == 0; return p
A, C = (alloc(n * n * 8 + (1 << 20)) for _ in range(2))
for b in (A, C):
    assert lib.ek_hip_synth_matrix_device(n, 1, b, n) == 0
sec = (ctypes.c_double * 3)()
for m in (16320, 12288, 8192, 5120):
    for swap in (0, 1):
        rc = lib.ek_hip_debug_corun(m, A, n, C, n, swap, sec); assert rc == 0, rc
        print("m=%5d symm on %s-priority stream: symm %.3f ms, update256 %.3f ms, together %.3f ms (sum %.3f)" % (m, "normal" if swap else "high", sec[0]*1e3, sec[1]*1e3, sec[2]*1e3, (sec[0]+sec[1])*1e3), flush=True)
