#!/usr/bin/env python3
"""Timing of the distributed tridiagonalisation on ONE GPU (tools, not product).

  python tools/team_timing.py N [P ...]

For each P: the whole team of P ranks rehearsed back to back on this GPU (ek_hip_debug_sytrd_team):
seconds / P is what one rank spends computing (its symv share, the replicated colupd, its share of
the trailing updates, yreduce) -- the RCCL latency per column is NOT in it (one GPU).  P = 0 runs
one rank over a size-1 RCCL communicator (adds the cost of issuing ncclAllReduce per column).
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigenkernel_amd import solver  # noqa: E402

n = int(sys.argv[1])
teams = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
lib = solver.load_library()
assert lib.ek_hip_init(0) == 0
sec = ctypes.c_double(0)
# the team form of the dense -> band stage (what a team runs from order 512 on: the stages after it are replicated
# -- bulge chasing -- or sharded by eigenvector columns)
ts = (ctypes.c_double * 4)(); flag = ctypes.c_int(0)
lib.ek_hip_debug_two_stage_timing(n, min(n, 1024), 1, ts, ctypes.byref(flag))
lib.ek_hip_debug_two_stage_timing(n, min(n, 1024), 2, ts, ctypes.byref(flag))
print("n=%d single GPU: dense->band %.4f s, band->tridiagonal %.4f s" % (n, ts[0], ts[1]), flush=True)
for P in teams:
    if P >= 1:
        assert lib.ek_hip_debug_sy2sb_team_timing(n, P, 1, ctypes.byref(sec)) == 0
        assert lib.ek_hip_debug_sy2sb_team_timing(n, P, 2, ctypes.byref(sec)) == 0
        print("n=%d team of %d rehearsed: dense->band %.4f s total, %.4f s per rank" % (n, P, sec.value, sec.value / P), flush=True)
if os.environ.get("EK_TEAM_TWO_STAGE_ONLY"):
    sys.exit(0)
assert lib.ek_hip_debug_sytrd(n, 0, 1, ctypes.byref(sec)) == 0
assert lib.ek_hip_debug_sytrd(n, 0, 2, ctypes.byref(sec)) == 0
print("n=%d single-GPU sytrd: %.4f s" % (n, sec.value), flush=True)
red = (ctypes.c_double * 2)()
for P in teams:
    if P >= 1:
        assert lib.ek_hip_debug_reduce_team(n, P, 1, red) == 0
        assert lib.ek_hip_debug_reduce_team(n, P, 2, red) == 0
        print("n=%d team of %d rehearsed: potrf %.4f s total (%.4f per rank), sygst %.4f s total (%.4f per rank)"
              % (n, P, red[0], red[0] / P, red[1], red[1] / P), flush=True)
for P in teams:
    if P == 0:
        solver.comm_init(solver.comm_unique_id(), 1, 0)
    assert lib.ek_hip_debug_sytrd_team(n, P, 1, ctypes.byref(sec)) == 0
    rc = lib.ek_hip_debug_sytrd_team(n, P, 2, ctypes.byref(sec))
    assert rc == 0, rc
    if P == 0:
        print("n=%d one rank over RCCL (world 1): %.4f s" % (n, sec.value), flush=True)
        solver.comm_destroy()
    else:
        print("n=%d team of %d rehearsed: %.4f s total, %.4f s per rank" % (n, P, sec.value, sec.value / P), flush=True)
