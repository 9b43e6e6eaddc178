#!/usr/bin/env python3
"""Randomised sweep of the process-grid modes: every rank of a random grid is played on the one
GPU (replicated inputs, or distributed inputs through the exchange hook) for random orders,
block sizes, n_vec and matrix kinds (incl. heavy deflation); the assembled pieces must equal
the 1x1 result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from eigenkernel_amd import solver, descriptor as d
from oracle import ek_oracle as ok
from test_host_logic import virtual_allgatherv

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
lib = solver.load_library(); lib.ek_hip_init(0)
bad = 0
for case in range(ncase):
    n = int(rng.integers(1, 700)) if rng.integers(0, 4) else int(rng.integers(700, 2400))
    kind = ["synth", "synth", "diag", "identity", "clustered", "block"][int(rng.integers(0, 6))]
    if kind == "synth": A = ok.synth_matrix(n, 1 + case % 7)
    elif kind == "diag": A = np.diag(rng.standard_normal(n))
    elif kind == "identity": A = np.eye(n) * 3.0
    elif kind == "clustered":
        Q, _ = np.linalg.qr(rng.standard_normal((n, n))); lam = np.repeat(rng.standard_normal(max(1, n // 7 + 1)), 7)[:n]
        A = (Q * lam) @ Q.T; A = (A + A.T) / 2
    else:
        A = np.zeros((n, n)); h = n // 2
        A[:h, :h] = ok.synth_matrix(h, 3) if h else 0; A[h:, h:] = ok.synth_matrix(n - h, 4)
    A = np.asfortranarray(A)
    gep = bool(rng.integers(0, 2))
    B = ok.synth_matrix(n, 9) if gep else None
    nprow, npcol = [(1, 2), (1, 3), (2, 2), (1, 8), (2, 4), (3, 1), (4, 2)][int(rng.integers(0, 7))]
    nb = int([1, 3, 16, 32, 64, 100][int(rng.integers(0, 6))])
    nv = int(rng.integers(0, n + 1)) if rng.integers(0, 2) else n
    inputs = "distributed" if rng.integers(0, 2) else "replicated"
    name = ("general_hip" if gep else "hip") + ("_select" if nv < n else "")
    ref, _ = solver.eigen_solver("general_hip" if gep else "hip", A, B)
    nbu = int(d.setup_distributed_matrix(n, n, nprow, npcol, 0, 0, block_size=nb)[0][d.BLOCK_ROW_])
    hook = virtual_allgatherv([A, B] if gep else [A], nbu, nprow, npcol)
    solver.set_allgatherv(hook if inputs == "distributed" else None)
    pieces = {}
    okv = True
    # half of the cases with the D&C's team form rehearsed by every cell (a team of the grid's columns, 1 - 3 heights
    # forced: small orders, heavy deflation, ragged strips): must change no bit
    team = npcol >= 2 and bool(rng.integers(0, 2))
    solver.stedc_team(npcol, int(rng.integers(1, 4))) if team else solver.stedc_team()
    for rank in range(nprow * npcol):
        hook.state["rank"] = rank
        myrow, mycol = rank // npcol, rank % npcol
        ep, _ = solver.eigen_solver(name, A, B, n_vec=nv if nv < n else None, block_size=nb,
                                    proc=solver.Process(rank, nprow * npcol, 0, nprow, npcol, myrow, mycol), inputs=inputs)
        okv = okv and np.array_equal(ep.values, ref.values)
        pieces[(myrow, mycol)] = ep.Vectors
    solver.set_allgatherv(None)
    solver.stedc_team()
    Zg = d.assemble_global(pieces, n, n, nbu, nprow, npcol)
    dz = float(np.abs(Zg[:, :nv] - ref.Vectors[:, :nv]).max()) if nv else 0.0
    flag = "" if (okv and dz <= 1e-13) else "  <-- BAD"
    bad += bool(flag)
    print("n=%4d %-9s %s grid %dx%d nb=%3d(%3d) nv=%4d %-11s dc_team=%d values_equal=%s max|dZ|=%.1e%s"
          % (n, kind, "GEP" if gep else "SEP", nprow, npcol, nb, nbu, nv, inputs, team, okv, dz, flag), flush=True)
print("BAD:", bad)
sys.exit(1 if bad else 0)
