"""GPU parity of the distributed tridiagonalisation (PDSYTRD on a 1 x P grid).

A pool box has one GPU and RCCL refuses two ranks on one device, so the algorithm is checked in
two halves that together make the production path:
  * the whole team of P ranks rehearsed inside one process (every rank with its own copy of A and
    its own workspace; the per-column exchange is a device kernel with all-reduce semantics):
    ownership of the strips, stale non-owned tiles, the travelling raw column, the per-strip
    trailing updates -- everything except the wire;
  * the RCCL binding itself (ncclCommInitRank / ncclAllReduce on the library's stream) with a
    communicator of size 1, through the same code path a rank of a larger team takes.
Results are held to the same bounds as the single-GPU stage (tests/test_gpu_blocks.py) and to
bit-identity across the ranks of a team.
"""
import os

import numpy as np
import pytest


def _free_port():
    """A TCP port nobody is listening on (127.0.0.1): fixed port numbers collide with sockets of earlier tests."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


pytestmark = pytest.mark.gpu
EPS = 2.220446049250313e-16


def _tridiag_check(A, Ar, d, e, tau, oracle):
    n = A.shape[0]
    T = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
    Q = oracle.ormtr_lower(Ar, tau, np.eye(n))
    return np.abs(Q @ T @ Q.T - A).max(), np.abs(Q.T @ Q - np.eye(n)).max()


def _check_against_single(hip, oracle, A, Ar, d, e, tau):
    n = A.shape[0]
    scale = np.abs(A).max()
    Ar1, d1, e1, tau1, info1 = hip.sytrd(A)
    assert info1 == 0
    # same algorithm, different summation order of y = A22 x: element-wise only loosely
    assert np.abs(d - d1).max() <= 1e-9 * scale
    assert np.abs(e - e1).max() <= 1e-9 * scale
    assert np.abs(tau - tau1).max() <= 1e-9
    res, orth = _tridiag_check(A, Ar, d, e, tau, oracle)
    assert res <= 64 * n * EPS * scale
    assert orth <= 64 * n * EPS
    T = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
    assert np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(A)).max() <= 8 * n * EPS * scale


@pytest.mark.parametrize("n,P", [(2, 2), (3, 4), (64, 2), (130, 2), (257, 3), (400, 4), (700, 2),
                                 (700, 8), (1000, 5), (1100, 16), (1537, 8)])
def test_sytrd_team_rehearsal(hip, oracle, n, P):
    A = oracle.synth_matrix(n, 1)
    Ar, d, e, tau, info, mismatch = hip.sytrd_team(A, P)
    assert info == 0
    assert mismatch == 0          # every rank of the team ends with the same bits
    _check_against_single(hip, oracle, A, Ar, d, e, tau)


def test_sytrd_team_of_one_is_the_same_algorithm(hip, oracle):
    """P = 1 through the distributed code path (exchange of a team of one is the identity)."""
    n = 515
    A = oracle.synth_matrix(n, 1)
    Ar, d, e, tau, info, mismatch = hip.sytrd_team(A, 1)
    assert info == 0 and mismatch == 0
    _check_against_single(hip, oracle, A, Ar, d, e, tau)


def test_sytrd_team_reproducible(hip, oracle):
    A = oracle.synth_matrix(600, 1)
    r1 = hip.sytrd_team(A, 4)
    r2 = hip.sytrd_team(A, 4)
    for a, b in zip(r1[:4], r2[:4]):
        assert np.array_equal(a, b)


def test_sytrd_team_reads_only_owned_strips(hip, oracle, monkeypatch):
    """NaN in every strip a rank does not own (what the distributed reduction to standard form
    leaves undefined there): the result must not change by a bit."""
    A = oracle.synth_matrix(900, 1)
    ref = hip.sytrd_team(A, 4)
    monkeypatch.setenv("EK_HIP_TEAM_POISON", "1")
    got = hip.sytrd_team(A, 4)
    assert got[4] == 0 and got[5] == 0
    for a, b in zip(ref[1:4], got[1:4]):
        assert np.array_equal(a, b)
    assert np.array_equal(np.tril(ref[0]), np.tril(got[0]))


def _band_similarity(A, Ab, V, tau):
    n = A.shape[0]
    Bw = 64
    L = np.tril(Ab) - np.tril(Ab, -(Bw + 1))
    Bd = L + np.tril(L, -1).T
    Q = np.eye(n)
    for j in range(n - 1, -1, -1):
        if tau[j] != 0.0:
            v = V[:, j]
            Q -= tau[j] * np.outer(v, v @ Q)
    return (np.linalg.norm(Q.T @ Q - np.eye(n)), np.linalg.norm(Q.T @ A @ Q - Bd) / np.linalg.norm(A),
            np.abs(np.linalg.eigvalsh(A) - np.linalg.eigvalsh(Bd)).max() / np.abs(np.linalg.eigvalsh(A)).max())


@pytest.mark.parametrize("n,P", [(66, 2), (130, 2), (200, 3), (257, 4), (449, 2), (700, 8), (1000, 5), (1000, 16), (1537, 3)])
def test_dense_to_band_team_rehearsal(hip, oracle, n, P):
    """The first stage of the two-stage tridiagonalisation over a 1 x P team (the distributed form of
    solver_scalapack_all.f90:59 for orders >= 512): the members' bands, reflectors and tau must be the same bits, and the
    result an orthogonal similarity to a band matrix at rounding level, like the single-GPU stage."""
    A = oracle.synth_matrix(n, 1)
    Ab, V, tau, flag, mism = hip.sy2sb_team(A, P)
    assert flag == 0 and mism == 0
    assert np.abs(np.tril(Ab, -65)).max() == 0.0
    orth, sim, ev = _band_similarity(A, Ab, V, tau)
    assert orth <= 64 * n * EPS and sim <= 32 * n * EPS and ev <= 4 * n * EPS


@pytest.mark.parametrize("kind", ["band65", "band4", "parallel_columns", "zero_column"])
@pytest.mark.parametrize("n,P", [(600, 3)])
def test_dense_to_band_team_rescues_panels(hip, oracle, kind, n, P):
    """Panels CholeskyQR2 cannot (or should not: condition beyond ~1e3) factor are factored by Householder reflections on
    the strip's owner and broadcast like any other panel: same bits on every member, an orthogonal similarity to a band."""
    rng = np.random.default_rng(n + P)
    if kind.startswith("band"):
        hw = int(kind[4:])
        M = np.tril(rng.standard_normal((n, n))); M = M - np.tril(M, -(hw + 1)); A = M + np.tril(M, -1).T
    else:
        A = oracle.synth_matrix(n, 1)
        if kind == "parallel_columns":
            A[64:, 5] = A[64:, 4] * (1.0 + 1e-7); A[5, 64:] = A[64:, 5]
            A[200:, 150] = A[200:, 149] + 1e-6 * rng.standard_normal(n - 200); A[150, 200:] = A[200:, 150]
        else:
            A[64:, 3] = 0.0; A[3, 64:] = 0.0
    Ab, V, tau, flag, mism = hip.sy2sb_team(A, P)
    assert flag & 0xff == 0 and mism == 0
    assert flag >> 8 >= 1                            # at least one panel took the rescue
    assert np.abs(np.tril(Ab, -65)).max() == 0.0
    orth, sim, ev = _band_similarity(A, Ab, V, tau)
    assert orth <= 64 * n * EPS and sim <= 32 * n * EPS and ev <= 4 * n * EPS


def test_dense_to_band_team_reads_only_owned_strips(hip, oracle, monkeypatch):
    """Every member's copy of the matrix is NaN outside its own 128-wide strips: the team form must neither read nor
    need them (what the distributed reduction to standard form leaves behind)."""
    n, P = 900, 3
    A = oracle.synth_matrix(n, 1)
    ref = hip.sy2sb_team(A, P)
    monkeypatch.setenv("EK_HIP_TEAM_POISON", "1")
    Ab, V, tau, flag, mism = hip.sy2sb_team(A, P)
    assert flag == 0 and mism == 0
    assert np.array_equal(Ab, ref[0]) and np.array_equal(V, ref[1]) and np.array_equal(tau, ref[2])
    assert np.isfinite(Ab).all() and np.isfinite(V).all()


def test_dense_to_band_team_of_one_over_rccl(hip, oracle, comm1):
    n = 700
    A = oracle.synth_matrix(n, 1)
    Ab, V, tau, flag, mism = hip.sy2sb_team(A, 0)
    assert flag == 0 and mism == 0
    orth, sim, ev = _band_similarity(A, Ab, V, tau)
    assert orth <= 64 * n * EPS and sim <= 32 * n * EPS and ev <= 4 * n * EPS


@pytest.mark.parametrize("n,P", [(5, 2), (128, 2), (130, 3), (300, 1), (640, 4), (1000, 8), (1000, 16), (1537, 5)])
def test_sygst_team_rehearsal(hip, oracle, n, P):
    """PDSYGST on a 1 x P grid: column-sharded solves + one all-gather; every strip is taken from
    its owner and the lower triangle held to the single-GPU stage's bound."""
    A = oracle.synth_matrix(n, 1)
    B = oracle.synth_matrix(n, 2)
    L, info = oracle.potrf_lower(B)
    assert info == 0
    L = np.tril(L)
    C_or = oracle.sygst_lower(A, L)
    got, info = hip.sygst_team(A, L, P)
    assert info == 0
    il = np.tril_indices(n)
    scale = np.abs(C_or[il]).max()
    assert np.abs(got[il] - C_or[il]).max() <= 32 * n * EPS * scale
    # oracle-independent identity: L C L^T = A
    Cs = np.tril(got) + np.tril(got, -1).T
    assert np.abs(L @ Cs @ L.T - A).max() <= 64 * n * EPS * np.abs(A).max()


@pytest.mark.parametrize("n,P", [(1, 2), (5, 3), (128, 2), (129, 2), (300, 1), (640, 4), (1000, 8), (1000, 16), (1537, 5)])
def test_potrf_team_rehearsal(hip, oracle, n, P):
    """PDPOTRF on a 1 x P grid: owner factors + one broadcast per block column; L and the block
    inverses complete and bit-identical on every rank."""
    B = oracle.synth_matrix(n, 2)
    L_or, info_or = oracle.potrf_lower(B)
    got, info, mismatch = hip.potrf_team(B, P)
    assert info == info_or == 0
    assert mismatch == 0
    il = np.tril_indices(n)
    assert np.abs(got[il] - L_or[il]).max() <= 16 * n * EPS * np.abs(L_or[il]).max()
    Lg = np.tril(got)
    assert np.abs(Lg @ Lg.T - B).max() <= 16 * n * EPS * np.abs(B).max()
    # the look-ahead (next strip's chain and broadcast on a second stream beside the rest of the update) changes no bit
    lib = hip.load_library()
    assert lib.ek_hip_debug_potrf_team_profile(0, 0) == 0
    try:
        off, info_off, mismatch_off = hip.potrf_team(B, P)
    finally:
        assert lib.ek_hip_debug_potrf_team_profile(-1, 0) == 0
    assert info_off == 0 and mismatch_off == 0 and np.array_equal(np.tril(off), np.tril(got))


def test_potrf_team_reports_the_failing_pivot_on_every_rank(hip, oracle):
    B = oracle.synth_matrix(400, 2)
    B[150, 150] = -1.0
    _, info_or = oracle.potrf_lower(B)
    _, info, mismatch = hip.potrf_team(B, 4)      # block column 1 belongs to rank 1, not to rank 0
    assert info == info_or == 151
    # the count includes a unit for every rank whose info differs from rank 0's
    _, info1, _ = hip.potrf_team(B, 1)
    assert info1 == 151


def test_potrf_over_rccl_world_of_one(hip, oracle, comm1):
    B = oracle.synth_matrix(700, 2)
    got, info, _ = hip.potrf_team(B, 0)
    ref, info1, _ = hip.potrf_team(B, 1)
    assert info == info1 == 0
    assert np.array_equal(np.tril(got), np.tril(ref))


def test_sygst_over_rccl_world_of_one(hip, oracle, comm1):
    n = 700
    A = oracle.synth_matrix(n, 1)
    L = np.tril(oracle.potrf_lower(oracle.synth_matrix(n, 2))[0])
    got, info = hip.sygst_team(A, L, 0)          # grouped ncclBroadcast all-gather, one rank
    assert info == 0
    ref, info1 = hip.sygst_team(A, L, 1)
    assert info1 == 0
    assert np.array_equal(got, ref)


def test_sytrd_team_rejects_bad_team(hip, oracle):
    A = oracle.synth_matrix(8, 1)
    assert hip.sytrd_team(A, 17)[4] == -7
    assert hip.sytrd_team(A, -1)[4] == -7


@pytest.fixture
def comm1(hip):
    """A size-1 RCCL communicator attached to the library (the binding a rank of a team uses)."""
    uid = hip.comm_unique_id()
    assert len(uid) == 128
    hip.comm_init(uid, 1, 0)
    lib = hip.load_library()
    assert lib.ek_hip_comm_size() == 1 and lib.ek_hip_comm_rank() == 0
    yield lib
    hip.comm_destroy()
    assert lib.ek_hip_comm_size() == 0 and lib.ek_hip_comm_rank() == -1


def test_rccl_allreduce_binding(hip, comm1):
    import ctypes
    lib = comm1
    x = np.arange(1000, dtype=np.float64) * 0.5 - 3.0
    dptr = ctypes.c_void_p()
    assert lib.ek_hip_malloc(ctypes.byref(dptr), x.nbytes) == 0
    try:
        assert lib.ek_hip_memcpy_h2d(dptr, x.ctypes.data_as(ctypes.c_void_p), x.nbytes) == 0
        assert lib.ek_hip_comm_allreduce_device(dptr, x.size) == 0
        y = np.zeros_like(x)
        assert lib.ek_hip_memcpy_d2h(y.ctypes.data_as(ctypes.c_void_p), dptr, x.nbytes) == 0
        assert np.array_equal(x, y)          # sum over one rank
    finally:
        lib.ek_hip_free(dptr)


def test_allreduce_without_communicator(hip):
    lib = hip.load_library()
    assert lib.ek_hip_comm_size() == 0
    assert lib.ek_hip_comm_allreduce_device(None, 0) == -995


@pytest.mark.parametrize("n", [130, 700])
def test_sytrd_over_rccl_world_of_one(hip, oracle, comm1, n):
    A = oracle.synth_matrix(n, 1)
    Ar, d, e, tau, info, mismatch = hip.sytrd_team(A, 0)     # 0: use the attached communicator
    assert info == 0
    _check_against_single(hip, oracle, A, Ar, d, e, tau)
    # and it is the very computation of the rehearsal with a team of one
    Ar1, d1, e1, tau1, info1, _ = hip.sytrd_team(A, 1)
    assert info1 == 0
    assert np.array_equal(d, d1) and np.array_equal(e, e1) and np.array_equal(tau, tau1)
    assert np.array_equal(np.tril(Ar), np.tril(Ar1))


def test_sytrd_team_zero_needs_communicator(hip, oracle):
    A = oracle.synth_matrix(8, 1)
    assert hip.sytrd_team(A, 0)[4] == -7


@pytest.mark.parametrize("two_stage", [False, True])
@pytest.mark.parametrize("min_ranks", ["3", "1"])
@pytest.mark.parametrize("problem,n,n_vec", [("gep", 300, 300), ("sep", 515, 515), ("gep", 400, 37)])
def test_whole_path_with_communicator_attached(hip, oracle, comm1, monkeypatch, min_ranks, problem, n, n_vec, two_stage):
    """ek_hip_solve_device_grid with a communicator of the grid's size takes the distributed
    tridiagonalisation (here 1 x 1 over RCCL) and must agree with the plain single-GPU solve.
    two_stage: the two-stage tridiagonalisation instead (the strips of the distributed reduction are
    all-gathered over the communicator, nothing is exchanged per column)."""
    import ctypes
    lib = comm1
    hip.set_two_stage(100 if two_stage else 0)
    # "1": the distributed Cholesky factor and reduction to standard form are taken as well
    monkeypatch.setenv("EK_HIP_DIST_MIN_RANKS", min_ranks)
    A = oracle.synth_matrix(n, 1)
    B = oracle.synth_matrix(n, 2) if problem == "gep" else None
    pb = 1 if problem == "gep" else 0

    def dev(M):
        p = ctypes.c_void_p()
        assert lib.ek_hip_malloc(ctypes.byref(p), M.nbytes) == 0
        assert lib.ek_hip_memcpy_h2d(p, M.ctypes.data_as(ctypes.c_void_p), M.nbytes) == 0
        return p

    def run(grid):
        dA = dev(np.asfortranarray(A)); dB = dev(np.asfortranarray(B)) if B is not None else None
        w = np.zeros(n); Z = np.zeros((n, n_vec), order="F")
        dw = dev(w); dZ = dev(Z)
        try:
            if grid:
                info = lib.ek_hip_solve_device_grid(pb, n, n_vec, dA, n, dB, n, dw, dZ, n, 64, 1, 1, 0, 0, None, 0)
            else:
                info = lib.ek_hip_solve_device(pb, n, n_vec, dA, n, dB, n, dw, dZ, n, None, 0)
            assert info == 0
            assert lib.ek_hip_memcpy_d2h(w.ctypes.data_as(ctypes.c_void_p), dw, w.nbytes) == 0
            assert lib.ek_hip_memcpy_d2h(Z.ctypes.data_as(ctypes.c_void_p), dZ, Z.nbytes) == 0
        finally:
            for p in (dA, dB, dw, dZ):
                if p is not None:
                    lib.ek_hip_free(p)
        return w, Z

    w_d, Z_d = run(True)
    w_or = oracle.solve(A, B)[0] if B is not None else np.linalg.eigvalsh(A)
    tol = 4 * n * EPS * np.abs(w_or).max()
    assert np.abs(w_d - w_or).max() <= tol
    Bm = B if B is not None else np.eye(n)
    R = A @ Z_d - (Bm @ Z_d) * w_d[:n_vec]
    assert np.abs(R).max() <= 1e-12
    G = Z_d.T @ Bm @ Z_d
    assert np.abs(G - np.eye(n_vec)).max() <= 1e-11
    hip.set_two_stage(-1)


# ------------------------------------------------------------------------------------------------
# Real multi-process runs on ONE GPU: three ranks (processes) share the device, each holds only its
# own data, and every exchange of the distributed stages travels between the processes through
# the host hook (gloo all-gather on CPU tensors) -- ek_hip_comm_attach_host.  What the team
# rehearsal cannot show (a rank other than 0 taking the one-member code path, true process
# separation) is shown here; what neither shows is RCCL's wire, which needs more than one GPU.
def _mp_worker(rank, world, port, q):
    import faulthandler
    import sys
    import traceback

    def say(what):
        sys.stderr.write("[rank %d] %s\n" % (rank, what)); sys.stderr.flush()
    try:
        say("start")
        import torch.distributed as dist      # the first import on a fresh box can take minutes
        faulthandler.dump_traceback_later(150, exit=True)     # from here a stuck rank says where, then goes away
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from eigenkernel_amd import solver as sv, descriptor as d
        from oracle import ek_oracle
        say("process group up")
        lib = sv.load_library()
        assert lib.ek_hip_init(0) == 0
        say("GPU bound")
        base_hook = sv.torch_allgatherv(dist)
        inject = {"calls": 0, "fail_at": -1}

        def hook(send, counts, displs):       # the exchange itself always happens; one call may then REPORT a failure
            res = base_hook(send, counts, displs)
            inject["calls"] += 1
            return res[:-1] if inject["calls"] == inject["fail_at"] else res
        hook.n_ranks = base_hook.n_ranks
        sv.set_allgatherv(hook)
        sv.comm_attach_host(world, rank)
        out = {}
        # (1) the distributed tridiagonalisation, this process being rank `rank` of the team
        n = 700
        A = ek_oracle.synth_matrix(n, 1)
        Ar, dd, ee, tau, info, _ = sv.sytrd_team(A, 0)
        assert info == 0
        out["sytrd"] = (dd.copy(), ee.copy(), tau.copy(), np.tril(Ar))
        say("sytrd done")
        # (2) Cholesky factor (complete on every rank) and the reduction (own strips only)
        B = ek_oracle.synth_matrix(n, 2)
        Lg, info, _ = sv.potrf_team(B, 0)
        assert info == 0
        out["potrf"] = np.tril(Lg)
        assert lib.ek_hip_debug_potrf_team_profile(0, 0) == 0      # the same without the look-ahead: same bits
        Lg0, info0, _ = sv.potrf_team(B, 0)
        assert lib.ek_hip_debug_potrf_team_profile(-1, 0) == 0
        assert info0 == 0 and np.array_equal(np.tril(Lg0), out["potrf"])
        C, info = sv.sygst_team(A, np.tril(Lg), 0)
        assert info == 0
        own = [c for c in range(n) if (c // 128) % world == rank]
        out["sygst_cols"] = (own, C[:, own].copy())
        say("potrf, sygst done")
        # (3) whole path, replicated inputs, 1 x world grid: this rank's block-cyclic piece of Z
        res = {}
        for solver_name, nv in (("general_hip", None), ("hip", None), ("general_hip_select", 50)):
            gen = solver_name.startswith("general")
            proc = sv.Process(rank, world, 0, 1, world, 0, rank)
            ep, _ = sv.eigen_solver(solver_name, A, B if gen else None, n_vec=nv, proc=proc)
            nb = int(ep.desc[d.BLOCK_ROW_])
            nvec = ep.n_vec
            cols = d.local_indices(nvec, nb, rank, world)
            res[solver_name] = (ep.values.copy(), cols, ep.Vectors[:, :len(cols)].copy())
            say(solver_name + " done")
        out["solve"] = res
        # (3b) the two-stream look-ahead of the team's dense -> band stage (panel chain + broadcast on a second stream beside
        # the update: from 1024 rows on) across real processes through the host communicator: same bits as without it
        n2 = 1500
        A2 = ek_oracle.synth_matrix(n2, 1)
        proc = sv.Process(rank, world, 0, 1, world, 0, rank)
        la = {}
        for key, val in (("on", None), ("off", "0")):
            if val is None:
                os.environ.pop("EK_SY2SB_DIST_LOOKAHEAD_MIN", None)
            else:
                os.environ["EK_SY2SB_DIST_LOOKAHEAD_MIN"] = val
            ep, _ = sv.eigen_solver("hip", A2, None, proc=proc)
            cols = d.local_indices(n2, int(ep.desc[d.BLOCK_ROW_]), rank, world)
            la[key] = (ep.values.copy(), ep.Vectors[:, :len(cols)].copy())
        os.environ.pop("EK_SY2SB_DIST_LOOKAHEAD_MIN", None)
        out["lookahead"] = la
        say("look-ahead on / off done")
        # (3c) the divide & conquer's team form (heights below the top merge sharded by strips of the compact bases, one
        # all-gather round per `world` strips: ek_stedc.hip) forced at this order, across real processes: same bits as
        # with those heights replicated -- for the full spectrum and for a *_select arm
        dc = {}
        for key, lv in (("team", 3), ("replicated", 0)):
            sv.stedc_team(0, lv)
            ep, _ = sv.eigen_solver("hip", A2, None, proc=proc)
            cols = d.local_indices(n2, int(ep.desc[d.BLOCK_ROW_]), rank, world)
            eps_, _ = sv.eigen_solver("general_hip_select", A, B, n_vec=90, proc=proc)
            dc[key] = (ep.values.copy(), ep.Vectors[:, :len(cols)].copy(), eps_.values.copy(), eps_.Vectors.copy())
        sv.stedc_team()
        out["dc_team"] = dc
        say("D&C team form on / off done")
        # (4) an exchange that fails on ONE rank in the middle of a solve (here: rank 1's hook reports a failure of its
        # 12th exchange from now, inside the dense -> band stage at this order) ends the call on EVERY rank with -996: the sticky
        # record travels in the team's votes (ek_comm.hip comm_vote); the next call starts clean
        if rank == 1:
            inject["fail_at"] = inject["calls"] + 12
        proc = sv.Process(rank, world, 0, 1, world, 0, rank)
        try:
            sv.eigen_solver("general_hip", A, B, proc=proc)
            out["info_exchange_failure"] = 0
        except sv.SolverError as exc:
            out["info_exchange_failure"] = exc.info
        inject["fail_at"] = -1
        ep, _ = sv.eigen_solver("general_hip", A, B, proc=proc)
        out["w_after_exchange_failure"] = ep.values.copy()
        say("exchange failure on one rank done")
        # (5) failures are reported alike on every rank: a B that is not positive definite (the
        # failing block column belongs to rank 1) and a NaN in A
        Bbad = B.copy(); Bbad[150, 150] = -1.0
        proc = sv.Process(rank, world, 0, 1, world, 0, rank)
        try:
            sv.eigen_solver("general_hip", A, Bbad, proc=proc)
            out["info_bad_B"] = 0
        except sv.SolverError as exc:
            out["info_bad_B"] = exc.info
        Anan = A.copy(); Anan[3, 2] = np.nan; Anan[2, 3] = np.nan
        try:
            sv.eigen_solver("general_hip", Anan, B, proc=proc)
            out["info_nan_A"] = 0
        except sv.SolverError as exc:
            out["info_nan_A"] = exc.info
        # and the team is still usable afterwards
        ep, _ = sv.eigen_solver("hip", A, proc=proc)
        out["w_after"] = ep.values.copy()
        sv.comm_destroy()
        q.put((rank, out, None))
        dist.barrier()
        dist.destroy_process_group()
        say("finished")
    except Exception:
        q.put((rank, None, traceback.format_exc()))
    faulthandler.cancel_dump_traceback_later()
    # hand the result over completely, then leave without the interpreter's tear-down (GPU runtime,
    # gloo threads): nothing a rank does after its report may hold up the test
    q.close(); q.join_thread()
    sys.stderr.flush()
    os._exit(0)


def test_three_processes_share_the_gpu_and_exchange_through_the_host(hip, oracle):
    # the standard library's spawn, not torch.multiprocessing: this (pytest) process has the ROCm
    # HIP runtime loaded through libek_hip.so and must not load PyTorch's bundled copy on top of it
    import multiprocessing as mp
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mp_worker, args=(r, world, port, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    import queue as _queue
    got = []
    try:
        for _ in procs:
            got.append(q.get(timeout=420 if not got else 200))
    except _queue.Empty:
        for p in procs:
            if p.is_alive():
                p.terminate()
        pytest.fail("only %d of %d ranks reported (their progress lines are on stderr)" % (len(got), world))
    res = sorted(got, key=lambda t: t[0])
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.terminate()
    for rank, out, err in res:
        assert err is None, "rank %d:\n%s" % (rank, err)
    outs = [r[1] for r in res]
    n = 700
    A = oracle.synth_matrix(n, 1)
    B = oracle.synth_matrix(n, 2)
    # sytrd: identical bits on the three processes, and the rehearsal of a team of three gives them too
    for o in outs[1:]:
        for a, b in zip(outs[0]["sytrd"], o["sytrd"]):
            assert np.array_equal(a, b)
    Ar, dd, ee, tau, info, mm = hip.sytrd_team(A, world)
    assert info == 0 and mm == 0
    assert np.array_equal(dd, outs[0]["sytrd"][0]) and np.array_equal(ee, outs[0]["sytrd"][1])
    assert np.array_equal(tau, outs[0]["sytrd"][2]) and np.array_equal(np.tril(Ar), outs[0]["sytrd"][3])
    _check_against_single(hip, oracle, A, outs[0]["sytrd"][3], *outs[0]["sytrd"][:3])
    for o in outs:
        assert o["info_exchange_failure"] == -996
        assert np.array_equal(o["w_after_exchange_failure"], outs[0]["solve"]["general_hip"][0])
        assert o["info_bad_B"] == 151 and o["info_nan_A"] == -4
        assert np.array_equal(o["w_after"], outs[0]["solve"]["hip"][0])
    # the team's two-stream look-ahead changes no bit, on any process
    for o in outs:
        assert np.array_equal(o["lookahead"]["on"][0], o["lookahead"]["off"][0])
        assert np.array_equal(o["lookahead"]["on"][1], o["lookahead"]["off"][1])
        assert np.array_equal(o["lookahead"]["on"][0], outs[0]["lookahead"]["on"][0])
    # nor does the divide & conquer's team form, and rank 0's eigenvalues are everybody's
    for o in outs:
        for a, b in zip(o["dc_team"]["team"], o["dc_team"]["replicated"]):
            assert np.array_equal(a, b)
        assert np.array_equal(o["dc_team"]["team"][0], outs[0]["lookahead"]["on"][0])
        assert np.array_equal(o["dc_team"]["team"][1], o["lookahead"]["on"][1])
    w2 = np.linalg.eigvalsh(oracle.synth_matrix(1500, 1))
    assert np.abs(outs[0]["lookahead"]["on"][0] - w2).max() <= 4 * 1500 * EPS * np.abs(w2).max()
    # potrf: the complete factor on every process
    for o in outs[1:]:
        assert np.array_equal(outs[0]["potrf"], o["potrf"])
    L = outs[0]["potrf"]
    assert np.abs(L @ L.T - B).max() <= 16 * n * EPS * np.abs(B).max()
    # sygst: every strip from its owner
    C = np.zeros((n, n))
    for o in outs:
        own, cols = o["sygst_cols"]
        C[:, own] = cols
    Cs = np.tril(C) + np.tril(C, -1).T
    assert np.abs(L @ Cs @ L.T - A).max() <= 64 * n * EPS * np.abs(A).max()
    # whole path: eigenvalues identical on all processes and equal to the oracle's; the pieces of Z
    # assemble to B-orthonormal eigenvectors with a small residual
    for name, Bm in (("general_hip", B), ("hip", None), ("general_hip_select", B)):
        w0 = outs[0]["solve"][name][0]
        for o in outs[1:]:
            assert np.array_equal(w0, o["solve"][name][0])
        w_or = oracle.solve(A, Bm)[0] if Bm is not None else np.linalg.eigvalsh(A)
        nvec = 50 if name.endswith("select") else n
        assert np.abs(w0[:nvec] - w_or[:nvec]).max() <= 4 * n * EPS * np.abs(w_or).max()
        Z = np.zeros((n, nvec))
        seen = np.zeros(nvec, dtype=int)
        for o in outs:
            _, cols, Zl = o["solve"][name]
            Z[:, cols] = Zl[:n, :]
            seen[cols] += 1
        assert (seen == 1).all()
        Bd = Bm if Bm is not None else np.eye(n)
        assert np.abs(A @ Z - (Bd @ Z) * w0[:nvec]).max() <= 1e-12
        assert np.abs(Z.T @ Bd @ Z - np.eye(nvec)).max() <= 1e-11


def _mp_grid_worker(rank, world, port, q, nprow, npcol, inputs, two_stage_min=None):
    """Whole path on an nprow x npcol grid, one process per cell, exchanges through the host."""
    import faulthandler
    import sys
    import traceback
    try:
        if two_stage_min is not None:
            os.environ["EK_HIP_TWO_STAGE_MIN"] = str(two_stage_min)
        import torch.distributed as dist
        faulthandler.dump_traceback_later(150, exit=True)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from eigenkernel_amd import solver as sv, descriptor as d
        from oracle import ek_oracle
        lib = sv.load_library()
        assert lib.ek_hip_init(0) == 0
        sv.set_allgatherv(sv.torch_allgatherv(dist))
        sv.comm_attach_host(world, rank)
        n = 450
        A = ek_oracle.synth_matrix(n, 1)
        B = ek_oracle.synth_matrix(n, 2)
        _, _, myrow, mycol = d.make_process_grid(rank, world, nprow, npcol)
        proc = sv.Process(rank, world, 0, nprow, npcol, myrow, mycol)
        ep, _ = sv.eigen_solver("general_hip", A, B, proc=proc, inputs=inputs)
        sys.stderr.write("[rank %d] (%d,%d) solved, stages %s\n" % (rank, myrow, mycol, sorted(ep.stage_seconds)[:1]))
        # the same team laid out as 1 x world: the cells of a process column split its eigenvector columns among them and
        # exchange row pieces at the end (ek_solve.hip, team_sendrecv), so both layouts must assemble to the same bits
        proc1 = sv.Process(rank, world, 0, 1, world, 0, rank)
        ep1, _ = sv.eigen_solver("general_hip", A, B, proc=proc1, inputs=inputs)
        out = (myrow, mycol, int(ep.desc[d.BLOCK_ROW_]), ep.values.copy(), ep.Vectors.copy(),
               getattr(ep, "B_loc", None), ep1.values.copy(), ep1.Vectors.copy(), int(ep1.desc[d.BLOCK_ROW_]))
        sv.comm_destroy()
        q.put((rank, out, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, None, traceback.format_exc()))
    faulthandler.cancel_dump_traceback_later()
    q.close(); q.join_thread()
    sys.stderr.flush()
    os._exit(0)


@pytest.mark.parametrize("inputs,two_stage_min", [("replicated", None), ("distributed", None), ("replicated", 100)])
def test_four_processes_on_a_2x2_grid(hip, oracle, inputs, two_stage_min):
    """The reference's near-square grid for four ranks (processes.f90:56-65): rank = myrow*npcol +
    mycol is the team index of the distributed stages whatever the grid's shape; with
    inputs="distributed" the ranks hand in block-cyclic pieces of A and B as the reference does and
    get back pieces of Z and of L.  two_stage_min = 100: the tridiagonalisation in two stages -- the
    team then completes the reduced matrix with one all-gather per round of strips and needs no
    exchange per Householder column at all."""
    import multiprocessing as mp
    import queue as _queue
    from eigenkernel_amd import descriptor as d
    world, nprow, npcol, n = 4, 2, 2, 450
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mp_grid_worker, args=(r, world, port, q, nprow, npcol, inputs, two_stage_min),
                         daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in procs:
            got.append(q.get(timeout=420 if not got else 200))
    except _queue.Empty:
        for p in procs:
            if p.is_alive():
                p.terminate()
        pytest.fail("only %d of %d ranks reported" % (len(got), world))
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.terminate()
    for rank, out, err in got:
        assert err is None, "rank %d:\n%s" % (rank, err)
    outs = [g[1] for g in sorted(got, key=lambda t: t[0])]
    A = oracle.synth_matrix(n, 1)
    B = oracle.synth_matrix(n, 2)
    w_or = oracle.solve(A, B)[0]
    nb = outs[0][2]
    for o in outs:
        assert o[2] == nb and np.array_equal(o[3], outs[0][3])
    w = outs[0][3]
    assert np.abs(w - w_or).max() <= 4 * n * EPS * np.abs(w_or).max()
    Z = d.assemble_global({(o[0], o[1]): o[4] for o in outs}, n, n, nb, nprow, npcol)
    assert np.abs(A @ Z - (B @ Z) * w).max() <= 1e-12
    assert np.abs(Z.T @ B @ Z - np.eye(n)).max() <= 1e-11
    # 2 x 2 against 1 x 4: the same team, the same bits
    Z1 = d.assemble_global({(0, r): o[7] for r, o in enumerate(outs)}, n, n, outs[0][8], 1, world)
    assert np.array_equal(outs[0][6], w) and np.array_equal(Z1, Z)
    if inputs == "distributed":      # B_loc came back as the pieces of L
        L = np.tril(d.assemble_global({(o[0], o[1]): o[5] for o in outs}, n, n, nb, nprow, npcol))
        assert np.abs(L @ L.T - B).max() <= 16 * n * EPS * np.abs(B).max()


@pytest.mark.parametrize("argv", [["1500", "2", "2", "1"], ["900", "1", "3", "0"], ["700", "2", "1", "1"]])
def test_plain_c_host_with_forked_ranks(hip, argv):
    """host/ek_ranks_demo.c: an MPI-shaped C program that uses nothing but include/ek_hip.h -- forked
    ranks on a process grid, the all-gather hook over shared memory, host communicator,
    ek_hip_solve on block-cyclic pieces; it checks its own eigenpairs and exits non-zero otherwise."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "host", "ek_ranks_demo")
    assert os.path.exists(exe), "build it with `make -C host` (or __graft_entry__.build())"
    out = subprocess.run(["timeout", "-k", "5", "120", exe] + argv, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "max |A z - lambda B z|" in out.stdout


MPIEXEC = "/opt/conda/bin/mpiexec"


@pytest.mark.parametrize("nranks,solver,n,extra,comm", [
    (4, "general_hip", 700, [], "hook"),          # 2 x 2 grid, the library borrows MPI_Allgatherv, runs replicated
    (4, "general_hip", 700, [], "host"),          # distributed stages, every exchange through the hook
    (4, "hip_select", 1000, ["-n", "60"], "host"),   # lowest 60 pairs (the test process holds the GPU too:
                                                     # at most 6 processes may, so no 2 x 3 grid here)
    (2, "general_hip_select", 500, ["-n", "17"], "host"),   # 1 x 2 grid
    (3, "hip", 333, ["--block-size", "15"], "hook"),  # 1 x 3 grid, ragged blocks
])
def test_fortran_mpi_host_on_a_process_grid(tmp_path, nranks, solver, n, extra, comm):
    """The reference's host shape (main.f90:29-35, 84-104; processes.f90:17-36, 56-65;
    distribute_matrix.f90:92-148) in its own language: host/eigenkernel_hip_mpi_app.f90 under mpiexec,
    several MPI ranks sharing the one GPU, each with the block-cyclic pieces of its grid cell.  Eigenvalues
    against the single-process Fortran host on the same synthetic input; residuals from the gathered Z."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "host", "eigenkernel_hip_mpi_app")
    one = os.path.join(root, "host", "eigenkernel_hip_app")
    assert os.path.exists(MPIEXEC), "MPICH (the oracle's ScaLAPACK path uses it too) is missing from this image"
    if not (os.path.exists(exe) and os.path.exists(one)):
        subprocess.check_call(["make", "-C", os.path.join(root, "host")])
    n_vec = int(extra[extra.index("-n") + 1]) if "-n" in extra else n
    env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCP", "HSA_TOOLS")) and k != "LD_PRELOAD"}
    ref = subprocess.run([one, "-s", solver, "--synthetic", str(n), "-o", "ev_one.dat"] + extra,
                         cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env)
    assert ref.returncode == 0, ref.stderr
    out = subprocess.run([MPIEXEC, "-np", str(nranks), exe, "-s", solver, "--synthetic", str(n), "--comm", comm,
                          "-c", "-1", "-o", "ev_mpi.dat", "-l", "log_mpi.json"] + extra,
                         cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    txt = out.stdout
    nprow = int((nranks + 1) ** 0.5)
    while nranks % nprow:
        nprow -= 1
    assert "BLACS process grid: %d x %d (%d)" % (nprow, nranks // nprow, nranks) in txt
    w1 = np.loadtxt(tmp_path / "ev_one.dat")[:, 1]
    w2 = np.loadtxt(tmp_path / "ev_mpi.dat")[:, 1]
    assert w1.shape == w2.shape == (n_vec,)
    assert np.abs(w1 - w2).max() <= n * EPS * np.abs(w1).max()
    spread = float([l for l in txt.splitlines() if l.startswith("eigenvalue spread across ranks:")][0].split(":")[1])
    assert spread == 0.0
    res_max = float([l for l in txt.splitlines() if l.startswith("residual norm (max):")][0].split(":")[1])
    assert res_max <= 2e-14
    import json
    log = json.load(open(tmp_path / "log_mpi.json"))
    assert log["n_procs"] == nranks and log["grid"] == [nprow, nranks // nprow] and log["comm"] == comm
    assert {"eigen_solver", "eigen_solver_scalapack_all:pdsytrd"} <= {e["name"] for e in log["events"]}


@pytest.mark.gpu
def test_bench_self_launch_rehearsed_on_one_gpu(hip):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two ranks itself (fresh children,
    no GPU call in the parent), the ranks share GPU 0 through the host communicator (--rehearse-on-one-gpu), the
    line says n_gpus = 2 and carries both the replicas measurement and the grid probe with its parity check."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(["timeout", "-k", "10", "420", sys.executable, os.path.join(root, "bench.py"), "--gpus", "2",
                          "--rehearse-on-one-gpu", "--order", "1536", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--no-host-path", "--no-other-configs"],
                         capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 1
    assert line["parity"]["residual_norm_max"] <= line["parity"]["bounds"]["residual_norm_max"]
    probe = line.get("grid_probe")
    assert probe is not None and not probe.get("error"), probe


@pytest.mark.parametrize("n,P,levels,n_vec,gen", [(1500, 3, 2, None, False), (2048, 8, 2, None, True), (1152, 2, 1, None, False),
                                                   (1500, 4, 3, 200, True), (1301, 5, 9, None, False)])
def test_stedc_team_rehearsal_changes_no_bit(hip, oracle, n, P, levels, n_vec, gen):
    """The team form of the divide & conquer (heights below the top merge cut into 128-wide strips of the compact bases,
    strip S on rank S mod P; ek_stedc.hip) rehearsed on one GPU: a grid cell plays every rank's strips in turn.  Each
    product is the same GEMM per output element as on one GPU, so eigenvalues and the cell's eigenvectors must be the
    replicated form's bit for bit -- on ragged orders (strips that straddle two merges), with more heights asked for
    than the tree has, and for a *_select arm."""
    A = oracle.synth_matrix(n, 1)
    B = oracle.synth_matrix(n, 2) if gen else None
    name = ("general_hip" if gen else "hip") + ("_select" if n_vec else "")
    for rank in sorted({0, P - 1, P // 2}):
        proc = hip.Process(rank, P, 0, 1, P, 0, rank)
        hip.stedc_team()
        ref, _ = hip.eigen_solver(name, A, B, n_vec=n_vec, proc=proc)
        hip.stedc_team(P, levels, profile=True)
        ep, _ = hip.eigen_solver(name, A, B, n_vec=n_vec, proc=proc)
        sec = hip.stedc_team_seconds()
        hip.stedc_team()
        assert np.array_equal(ep.values, ref.values)
        assert np.array_equal(ep.Vectors, ref.Vectors)
        assert sec[0] > 0.0 and sec[1] > 0.0 and sec[2] <= sec[1] <= sec[0]      # (the team form did run)
    w_or = oracle.solve(A, B)[0] if gen else np.linalg.eigvalsh(A)
    assert np.abs(ref.values - w_or[:len(ref.values)]).max() <= 4 * n * EPS * np.abs(w_or).max()


def _mp_ragged_pair_worker(rank, world, port, q):
    """Two ranks, a ragged order: rank 0's share of the eigenvector columns is more than half of them (no compact D&C bases
    there), rank 1's is not."""
    import faulthandler
    import sys
    import traceback
    try:
        import torch.distributed as dist
        faulthandler.dump_traceback_later(150, exit=True)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from eigenkernel_amd import solver as sv, descriptor as d
        from oracle import ek_oracle
        lib = sv.load_library()
        assert lib.ek_hip_init(0) == 0
        sv.set_allgatherv(sv.torch_allgatherv(dist))
        sv.comm_attach_host(world, rank)
        n = 1111
        A = ek_oracle.synth_matrix(n, 1)
        proc = sv.Process(rank, world, 0, 1, world, 0, rank)
        sv.stedc_team(0, 2)                      # the team form asked for at this order
        ep, _ = sv.eigen_solver("hip", A, None, proc=proc)
        sv.stedc_team()
        cols = d.local_indices(n, int(ep.desc[d.BLOCK_ROW_]), rank, world)
        out = (ep.values.copy(), cols, ep.Vectors[:, :len(cols)].copy())
        sv.comm_destroy()
        q.put((rank, out, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, None, traceback.format_exc()))
    faulthandler.cancel_dump_traceback_later()
    q.close(); q.join_thread()
    sys.stderr.flush()
    os._exit(0)


def test_two_ranks_on_a_ragged_order_take_the_same_decision_about_the_team_form(hip, oracle):
    """The D&C's team form needs the compact bases (a rank's columns at most half of them).  With two ranks and 64-wide blocks
    dealt round robin, order 1111 gives rank 0 576 columns (> 556) and rank 1 535: the decision has to be the team's, not the
    rank's, or rank 1 waits in an all-gather that rank 0 never enters (ek_solve.hip: all_compact)."""
    import multiprocessing as mp
    import queue as _queue
    world, n = 2, 1111
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mp_ragged_pair_worker, args=(r, world, port, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in procs:
            got.append(q.get(timeout=300 if not got else 120))
    except _queue.Empty:
        for p in procs:
            if p.is_alive():
                p.terminate()
        pytest.fail("only %d of %d ranks reported: the ranks did not agree about the team form" % (len(got), world))
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.terminate()
    for rank, out, err in got:
        assert err is None, "rank %d:\n%s" % (rank, err)
    outs = [g[1] for g in sorted(got, key=lambda t: t[0])]
    A = oracle.synth_matrix(n, 1)
    w = np.linalg.eigvalsh(A)
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.abs(outs[0][0] - w).max() <= 4 * n * EPS * np.abs(w).max()
    Z = np.zeros((n, n))
    for _, cols, Zl in outs:
        Z[:, cols] = Zl[:n, :]
    assert np.abs(A @ Z - Z * outs[0][0]).max() <= 1e-12
    assert np.abs(Z.T @ Z - np.eye(n)).max() <= 1e-11
