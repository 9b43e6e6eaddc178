import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from eigenkernel_amd import solver
lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
sec = ctypes.c_double(0)
for n in (1024, 2048, 4096):
    lib.ek_hip_debug_sytrd(n, 0, 1, ctypes.byref(sec))
    lib.ek_hip_debug_sytrd(n, 0, 3, ctypes.byref(sec))
    print("n=%d sytrd %.4f s = %.2f us per column" % (n, sec.value, 1e6 * sec.value / n), flush=True)
