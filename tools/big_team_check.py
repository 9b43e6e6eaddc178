import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from scipy.linalg import eigvalsh_tridiagonal
from eigenkernel_amd import solver as hip
from oracle import ek_oracle
lib = hip.load_library(); assert lib.ek_hip_init(0) == 0
n, P = 16384, 8
t0 = time.time(); A = ek_oracle.synth_matrix(n, 1); print("synth %.1f s" % (time.time() - t0), flush=True)
Ar, d, e, tau, info, mm = hip.sytrd_team(A, P); print("team info", info, "mismatch", mm, flush=True)
Ar1, d1, e1, tau1, info1 = hip.sytrd(A); print("single info", info1, flush=True)
w = eigvalsh_tridiagonal(d, e); w1 = eigvalsh_tridiagonal(d1, e1)
EPS = 2.220446049250313e-16
print("max |dlambda| %.3e  bound %.3e" % (np.abs(w - w1).max(), 8 * n * EPS * np.abs(w1).max()))
print("max |d-d1| %.3e max |e|-|e1| %.3e" % (np.abs(d - d1).max(), np.abs(np.abs(e) - np.abs(e1)).max()))
