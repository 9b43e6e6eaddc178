"""Whole-path solves at orders between the old and the new crossover of the two-stage tridiagonalisation (odd
orders included), accepted by the reference's residual / orthogonality quantities on the GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from eigenkernel_amd import solver
import test_gpu_configs as tc
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
for n, gep, nv in ((5120, True, 5120), (6001, False, 6001), (7777, True, 300), (5121, False, 5121), (9999, True, 9999)):
    with tc._Dev(lib) as dev:
        r = tc._solve_1x1(lib, dev, gep, n, nv)
        mx, orth = tc._acceptance(lib, gep, n, nv, r["dA0"], r["dB0"], r["dw"], r["dZ"])
        print("n=%d gep=%d n_vec=%d residual max %.2e orth %.2e  stages %s" % (n, gep, nv, mx, orth, np.round(r["stages"][:7], 4)), flush=True)
print("SANITY OK")
