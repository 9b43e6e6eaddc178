"""Larger ragged orders through the whole path (device arrays), accepted by the reference's quantities on the GPU."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from eigenkernel_amd import solver
import test_gpu_configs as tc
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
bad = 0
for n, gep, nv in ((12345, True, 12345), (10001, False, 5000), (16383, True, 16383), (16385, False, 700), (8191, True, 4096), (20011, True, 64), (13000, False, 13000)):
    with tc._Dev(lib) as dev:
        r = tc._solve_1x1(lib, dev, gep, n, nv)
        mx, orth = tc._acceptance(lib, gep, n, nv, r["dA0"], r["dB0"], r["dw"], r["dZ"])
        ok = mx <= 1e-14 * max(1.0, (n / 1024.0) ** 0.5) and orth <= 1e-11
        bad += 0 if ok else 1
        print("n=%d gep=%d n_vec=%d residual max %.2e orth %.2e %s stages %s" % (n, gep, nv, mx, orth, "ok" if ok else "BAD", np.round(r["stages"][:7], 4)), flush=True)
print("BAD:", bad)
