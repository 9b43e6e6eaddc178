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
# A rank's time in the dense -> band stage is NOT (whole team back to back) / P: the panel chain runs on ONE rank while
# the others wait for its broadcast, so it counts in full.  The serial pieces are measured with the look-ahead off (HIP
# events around every chain and every "rest of the update" section), then a rank's critical path is
#   without look-ahead : chains + (everything else) / P
#   with look-ahead    : first chain + sum_p max(chain p+1, update p / P) + (everything else but chains and updates) / P
# (the wire -- one broadcast and one all-reduce per panel -- is in neither: one GPU).
d2b = {}
for P in teams:
    if P >= 1:
        parts = (ctypes.c_double * 4)()
        assert lib.ek_hip_debug_sy2sb_team_profile(n, P, 1, 0, ctypes.byref(sec), parts) == 0
        assert lib.ek_hip_debug_sy2sb_team_profile(n, P, 2, 0, ctypes.byref(sec), parts) == 0
        T, C, U, M = parts[0], parts[1], parts[2], parts[3]
        serial = C + (T - C) / P
        lookahead = M + (T - C - U) / P
        assert lib.ek_hip_debug_sy2sb_team_profile(n, P, 2, -1, ctypes.byref(sec), None) == 0
        d2b[P] = (serial, lookahead)
        print("n=%d team of %d rehearsed: dense->band %.4f s back to back (chains %.4f, rest-of-update sections %.4f); "
              "per rank: %.4f s without look-ahead, %.4f s with (model); the rehearsal with look-ahead on takes %.4f s"
              % (n, P, T, C, U, serial, lookahead, sec.value), flush=True)
# what a rank of a 1 x P team computes per solve (generalized problem, full spectrum), stage by stage: the distributed
# stages rehearsed above / below (seconds / P), the replicated bulge chasing, and the column-sharded stages measured by
# playing one grid cell of the replicated-input mode (ek_hip_solve_device_grid: no exchange, so a cell's stage times do
# not depend on the others).  The wire (two exchanges per panel, the all-gathers) is NOT in it.
def dmalloc(nbytes):
    ptr = ctypes.c_void_p()
    assert lib.ek_hip_malloc(ctypes.byref(ptr), int(nbytes)) == 0
    return ptr


dA, dB, dw = dmalloc(n * n * 8), dmalloc(n * n * 8), dmalloc(n * 8)
red = (ctypes.c_double * 2)()
stage = (ctypes.c_double * 8)()
for P in teams:
    if P < 2:
        continue
    # the team's Cholesky factorisation like its dense -> band stage: the owner's chain of a strip (factor + invert the
    # diagonal block, solve the panel) runs on ONE rank while the others wait for its broadcast, so it counts in full --
    # measured with the look-ahead off (events around chains and rest-of-update sections), then priced with it
    pp = (ctypes.c_double * 4)()
    assert lib.ek_hip_debug_potrf_team_profile(0, 0) == 0
    assert lib.ek_hip_debug_reduce_team(n, P, 1, red) == 0
    assert lib.ek_hip_debug_potrf_team_profile(0, 1) == 0
    assert lib.ek_hip_debug_reduce_team(n, P, 1, red) == 0
    assert lib.ek_hip_debug_potrf_team_profile_get(P, pp) == 0
    T, C, U, M = red[0], pp[0], pp[1], pp[2]
    potrf_serial, potrf_la = C + (T - C) / P, M + (T - C - U) / P
    assert lib.ek_hip_debug_potrf_team_profile(-1, 0) == 0
    assert lib.ek_hip_debug_reduce_team(n, P, 2, red) == 0
    print("n=%d team of %d rehearsed: potrf %.4f s back to back without look-ahead (chains %.4f, rest-of-update sections %.4f); per rank: "
          "%.4f s without look-ahead, %.4f s with (model); the rehearsal with look-ahead on takes %.4f s; sygst %.4f s (%.4f per rank)"
          % (n, P, T, C, U, potrf_serial, potrf_la, red[0], red[1], red[1] / P), flush=True)
    ncl = (n + P - 1) // P
    dZ = dmalloc((ncl + 64) * n * 8)
    # the D&C's team form (heights below the top merge sharded by strips) rehearsed inside the cell's solve: the cell plays
    # every rank's sections in turn; a rank of a real team computes for (stage - all sections + the longest rank's) seconds
    dc = (ctypes.c_double * 3)()
    levels = int(os.environ.get("EK_TEAM_DC_LEVELS", "-1"))
    for rep in range(2):
        assert lib.ek_hip_synth_matrix_device(n, 1, dA, n) == 0
        assert lib.ek_hip_synth_matrix_device(n, 2, dB, n) == 0
        assert lib.ek_hip_debug_stedc_team(P, levels, 1) == 0
        rc = lib.ek_hip_solve_device_grid(1, n, n, dA, n, dB, n, dw, dZ, n, 64, 1, P, 0, 0, stage, 8)
        assert rc == 0, rc
        assert lib.ek_hip_debug_stedc_team_get(dc) == 0
        assert lib.ek_hip_debug_stedc_team(0, -1, 0) == 0
    stedc_rank = stage[4] - dc[1] + dc[2]
    print("n=%d P=%d D&C: stage %.4f s with every rank's sections back to back (%.4f s of sections, the longest rank's %.4f) -> %.4f s per rank"
          % (n, P, stage[4], dc[1], dc[2], stedc_rank), flush=True)
    parts = {"potrf (team, look-ahead, chain in full)": potrf_la, "sygst (team)": red[1] / P, "dense->band (team, look-ahead, chain in full)": d2b[P][1],
             "band->tridiagonal (replicated)": ts[1], "stedc (team form below the top merge, top merge on own columns)": stedc_rank,
             "Q2 + Q1 (own columns)": stage[5], "recovery (own columns)": stage[6]}
    print("n=%d P=%d per-rank compute: %.3f s = %s   [dense->band without look-ahead: %.3f]" % (n, P, sum(parts.values()),
          ", ".join("%s %.3f" % kv for kv in parts.items()), d2b[P][0]), flush=True)
    lib.ek_hip_free(dZ)
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
