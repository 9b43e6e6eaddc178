import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from eigenkernel_amd import solver
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
red = (ctypes.c_double * 2)()
for n in (4096, 8192, 16384):
    for P in (1, 2):
        assert lib.ek_hip_debug_reduce_team(n, P, 1, red) == 0
        assert lib.ek_hip_debug_reduce_team(n, P, 2, red) == 0
        print("n=%d team %d: potrf_dist %.4f s total, sygst_dist %.4f s total" % (n, P, red[0], red[1]), flush=True)
