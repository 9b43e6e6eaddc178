"""CPU suite: host-side mirror of the reference interface (descriptors, grid rules,
MatrixMarket I/O, result file format, verifier), the C-ABI export check and the N>1 bench
aggregation under gloo.  No GPU, no compute calls into libek_hip.so."""
import ctypes
import os
import json
import sys
import re

import numpy as np
import pytest


def _free_port():
    """A TCP port nobody is listening on (127.0.0.1): fixed port numbers collide with sockets of earlier tests."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


import eigenkernel_amd as ek
from eigenkernel_amd import descriptor as dsc
from eigenkernel_amd import solver
from eigenkernel_amd.matrix_io import _fortran_e26

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_procs_matches_reference_rule():
    """processes.f90:56-65; SURVEY.md section 5: 1->1x1, 2->1x2, 4->2x2, 8->2x4."""
    assert dsc.layout_procs(1) == (1, 1)
    assert dsc.layout_procs(2) == (1, 2)
    assert dsc.layout_procs(4) == (2, 2)
    assert dsc.layout_procs(6) == (2, 3)
    assert dsc.layout_procs(8) == (2, 4)
    assert dsc.layout_procs(7) == (1, 7)


def test_numroc_partitions_dimension():
    for n, nb, p in [(30, 15, 2), (400, 64, 2), (16384, 64, 4), (1000, 64, 3), (5, 64, 2)]:
        assert sum(dsc.numroc(n, nb, i, 0, p) for i in range(p)) == n
    assert dsc.numroc(30, 15, 0, 0, 2) == 15 and dsc.numroc(30, 15, 1, 0, 2) == 15
    # 400 = 6 full blocks of 64 + 16: blocks 0,2,4,(6: 16 rows) -> proc 0; blocks 1,3,5 -> proc 1
    assert dsc.numroc(400, 64, 0, 0, 2) == 208 and dsc.numroc(400, 64, 1, 0, 2) == 192


def test_setup_distributed_matrix_block_shrink_rule():
    """distribute_matrix.f90:114-120: NB shrinks to max(min(rows/P_r, cols/P_c), 1);
    BNZ30 on 2x2 -> NB = 15."""
    desc, mat = dsc.setup_distributed_matrix(30, 30, nprow=2, npcol=2, myrow=0, mycol=0)
    assert desc[dsc.BLOCK_ROW_] == 15 and desc[dsc.BLOCK_COL_] == 15
    assert mat.shape == (15, 15) and mat.flags.f_contiguous and not mat.any()
    desc, mat = dsc.setup_distributed_matrix(400, 400)
    assert list(desc) == [1, 0, 400, 400, 64, 64, 0, 0, 400]
    desc, _ = dsc.setup_distributed_matrix(400, 400, block_size=25)
    assert desc[dsc.BLOCK_ROW_] == 25
    desc, mat = dsc.setup_distributed_matrix(3, 3, nprow=2, npcol=4, myrow=1, mycol=3)
    assert desc[dsc.BLOCK_ROW_] == 1 and desc[dsc.LOCAL_ROWS_] >= 1


def test_matrix_market_roundtrip(tmp_path, golden_dir):
    A = ek.read_matrix_file(os.path.join(golden_dir, "ELSES_MATRIX_BNZ30_A.mtx"))
    assert A.size == 30 and A.num_non_zeros == 303 and A.suffix.shape == (2, 303)
    D = A.to_dense()
    assert np.array_equal(D, D.T)
    p = tmp_path / "a.mtx"
    ek.write_matrix_file(str(p), D)
    assert np.array_equal(ek.read_matrix_file(str(p)).to_dense(), D)
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(ValueError):
        ek.read_matrix_file(str(bad))


def test_eigenvalues_file_format_matches_reference_golden(tmp_path, golden_dir):
    """main.f90:113-118: format (I8, ' ', E26.16e3); the shipped _ev.txt is such a file."""
    lines = open(os.path.join(golden_dir, "ELSES_MATRIX_BNZ30_ev.txt")).read().splitlines()
    vals = [float(l.split()[1]) for l in lines]
    p = tmp_path / "eigenvalues.dat"
    ek.write_eigenvalues(str(p), vals)
    assert p.read_text().splitlines() == lines
    assert _fortran_e26(0.0).strip() == "0.0000000000000000E+000"


def test_verifier_normalisations():
    """verifier.f90:198-199 (avg = sum/||A||_F/n, max), :310-325 (scale by 1/sqrt(G_jj), zero diag)."""
    from eigenkernel_amd.verifier import eval_orthogonality, eval_residual_norm, get_ipratios
    rng = np.random.default_rng(0)
    n = 12
    A = rng.normal(size=(n, n)); A = A + A.T
    w, V = np.linalg.eigh(A)
    a_norm, ave, mx = eval_residual_norm(A, w, V)
    assert abs(a_norm - np.linalg.norm(A)) < 1e-12 and mx < 1e-14 and ave <= mx
    assert eval_orthogonality(V) < 1e-14
    assert eval_orthogonality(3.0 * V) < 1e-14          # column scaling is normalised away
    V2 = V.copy(); V2[:, 1] = V2[:, 0]
    assert eval_orthogonality(V2) > 1.0
    assert np.allclose(get_ipratios(np.eye(n)), 1.0)
    assert np.allclose(get_ipratios(np.ones((n, 2)) / np.sqrt(n)), 1.0 / n)


def test_header_and_library_export_the_same_symbols():
    """Every function include/ek_hip.h (the drop-in boundary) and include/ek_hip_debug.h (tuning / test
    hooks) declare is exported by libek_hip.so and bound by the host mirror (no compute call: works
    without a GPU); the boundary header carries no debug or profile hook."""
    hdr = open(os.path.join(ROOT, "include", "ek_hip.h")).read()
    dbg = open(os.path.join(ROOT, "include", "ek_hip_debug.h")).read()
    boundary = set(re.findall(r"\b(ek_hip_\w+)\s*\(", hdr))
    hooks = set(re.findall(r"\b(ek_hip_\w+)\s*\(", dbg))
    assert not any("debug" in f or "profile" in f for f in boundary)
    assert not (boundary & hooks)
    declared = boundary | hooks
    assert declared == set(solver.EXPORTED_SYMBOLS)
    assert os.path.exists(solver.LIB_PATH), "build with __graft_entry__.build()"
    lib = ctypes.CDLL(solver.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.ek_hip_stage_name.restype = ctypes.c_char_p
    names = [lib.ek_hip_stage_name(i).decode() for i in range(8)]
    # the reference's add_event names (generalized_to_standard.f90:33,44,111; solver_scalapack_all.f90:66,93,104,122)
    assert names[:7] == ["reduce_generalized:pdpotrf", "reduce_generalized:pdsygst",
                         "eigen_solver_scalapack_all:pdsytrd", "eigen_solver_scalapack_all:gather1",
                         "eigen_solver_scalapack_all:pdstedc", "eigen_solver_scalapack_all:pdormtr",
                         "recovery_generalized"]
    assert lib.ek_hip_version() >= 2


def test_argument_validation_without_gpu():
    """LAPACK-style info = -k for bad arguments is decided before any device work."""
    lib = solver.load_library()
    desc = dsc.descinit(4, 4, 4, 4, 0, 0, 0, 4)
    a = np.zeros((4, 4), order="F")
    ip = desc.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    dp = a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert lib.ek_hip_potrf(-1, dp, ip) == -1
    assert lib.ek_hip_potrf(4, None, ip) == -2
    bad = desc.copy(); bad[2] = 5
    assert lib.ek_hip_potrf(4, dp, bad.ctypes.data_as(ctypes.POINTER(ctypes.c_int))) == -303
    assert lib.ek_hip_solve(2, 4, 4, dp, ip, dp, ip, dp, dp, ip, 1, 1, 0, 0, None, 0) == -1
    assert lib.ek_hip_solve(1, 4, 5, dp, ip, dp, ip, dp, dp, ip, 1, 1, 0, 0, None, 0) == -3
    assert lib.ek_hip_solve(1, 4, 4, dp, ip, dp, ip, dp, dp, ip, 2, 1, 0, 0, None, 0) == -11
    assert lib.ek_hip_dgemm(0, 0, 4, 4, -1, 1.0, dp, 4, dp, 4, 0.0, dp, 4, 0) == -5
    # replicated-input mode for larger grids: full A (lda), Z by descriptor, grid cell checked
    dz = dsc.descinit(4, 4, 2, 2, 0, 0, 0, 2)
    iz = dz.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    rep = lib.ek_hip_solve_replicated
    assert rep(0, 4, 4, dp, 3, None, 4, dp, dp, iz, 2, 2, 0, 0, None, 0) == -5
    assert rep(1, 4, 4, dp, 4, None, 4, dp, dp, iz, 2, 2, 0, 0, None, 0) == -6
    assert rep(0, 4, 4, dp, 4, None, 4, dp, dp, iz, 0, 2, 0, 0, None, 0) == -11
    assert rep(0, 4, 4, dp, 4, None, 4, dp, dp, iz, 2, 2, 2, 0, None, 0) == -13
    assert rep(0, 4, 4, dp, 4, None, 4, dp, dp, iz, 2, 2, 0, -1, None, 0) == -14
    dz1 = dz.copy(); dz1[8] = 1                   # lld below the local row count of cell (0,*)
    assert rep(0, 4, 4, dp, 4, None, 4, dp, dp, dz1.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
               2, 2, 0, 0, None, 0) == -1009
    grid = lib.ek_hip_solve_device_grid
    assert grid(0, 4, 4, None, 4, None, 4, None, None, 4, 2, 1, 2, 0, 0, None, 0) == -4
    assert grid(0, 4, 4, 1, 4, None, 4, 1, 1, 4, 0, 1, 2, 0, 0, None, 0) == -11
    assert grid(0, 4, 4, 1, 4, None, 4, 1, 1, 1, 2, 1, 2, 0, 0, None, 0) == -10


def test_solver_dispatch_rejects_unknown_and_missing_library(tmp_path):
    with pytest.raises(ValueError):
        solver.eigen_solver("general_elpa2", np.eye(3), np.eye(3))
    with pytest.raises(solver.LibraryMissing):
        solver.load_library(str(tmp_path / "nope.so"))     # product path fails loudly


def _bench_agg_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eigenkernel_amd.parallel import aggregate_throughput, shard_problems
    mine = shard_problems(5, rank, world)
    value, t = aggregate_throughput(units_local=len(mine) * 100, seconds_local=1.0 + rank, dist=dist)
    q.put((rank, mine, value, t))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_rank_aggregation_gloo():
    """bench.py --gpus N: independent problems are sharded over ranks with no data-path
    collective; value = all units / max-over-ranks time (world_size 2, gloo)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_agg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    for _, _, value, t in res:
        assert t == 2.0 and abs(value - 500 / 2.0) < 1e-12


def test_block_cyclic_maps_partition_and_assemble():
    """Ownership rule of distribute_matrix.f90:128-138: global (i, j) lives on
    ((i / NB) mod P_r, (j / NB) mod P_c); local pieces tile the global matrix exactly once."""
    rng = np.random.default_rng(3)
    for n, nb, (pr, pc) in [(30, 15, (2, 2)), (100, 64, (1, 4)), (257, 32, (2, 4)), (7, 3, (3, 2))]:
        G = rng.standard_normal((n, n))
        seen = np.zeros(n, dtype=int)
        for p in range(pc):
            idx = dsc.local_indices(n, nb, p, pc)
            assert len(idx) == dsc.numroc(n, nb, p, 0, pc)
            assert all((g // nb) % pc == p for g in idx)
            seen[idx] += 1
        assert (seen == 1).all()
        pieces = {}
        for rank in range(pr * pc):
            _, _, myrow, mycol = dsc.make_process_grid(rank, pr * pc, pr, pc)
            assert rank == myrow * pc + mycol                    # row-major (processes.f90:23)
            ri = dsc.local_indices(n, nb, myrow, pr); ci = dsc.local_indices(n, nb, mycol, pc)
            pieces[(myrow, mycol)] = G[np.ix_(ri, ci)]
        assert np.array_equal(dsc.assemble_global(pieces, n, n, nb, pr, pc), G)
    assert dsc.make_process_grid(5, 8)[:2] == dsc.layout_procs(8) == (2, 4)
    with pytest.raises(ValueError):
        dsc.make_process_grid(0, 6, 4, 2)


def _grid_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eigenkernel_amd import descriptor as dsc
    n, n_vec, nb = 100, 37, 8
    nprow, npcol, myrow, mycol = dsc.make_process_grid(rank, world, 1, world)   # the 1 x P grid of bench.py's columns mode
    cols = dsc.local_indices(n_vec, nb, mycol, npcol)
    out = [None] * world
    dist.all_gather_object(out, (nprow, npcol, myrow, mycol, cols.tolist()))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_column_sharding_of_eigenvectors_gloo():
    """Replicated-input mode on a 1 x P grid: the ranks' eigenvector columns partition
    [0, n_vec) with no overlap (world_size 2, gloo)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]
    cells = res[0][1]
    assert [(c[0], c[1], c[2], c[3]) for c in cells] == [(1, 2, 0, 0), (1, 2, 0, 1)]
    allcols = sorted(cells[0][4] + cells[1][4])
    assert allcols == list(range(37))


def test_eigenvector_writer_formats(tmp_path):
    """matrix_io.f90:173-285: `<dir>/%08d.dat`, text `i j value` lines or one Fortran
    unformatted record (248 bytes for N = 30)."""
    assert ek.parse_printed_vecs_ranges("1-3,7") == [(1, 3), (7, 7)]
    with pytest.raises(ValueError):
        ek.parse_printed_vecs_ranges("-3")
    Z = np.arange(1, 31 * 30 + 1, dtype=np.float64).reshape(31, 30)[:30].T.copy() / 7.0
    ek.write_eigenvectors(str(tmp_path), Z, [(2, 3)], binary=False)
    lines = (tmp_path / "00000002.dat").read_text().splitlines()
    assert len(lines) == 30 and lines[0][:18] == "       1        2 "
    assert abs(float(lines[4].split()[2]) - Z[4, 1]) <= 1e-15 * abs(Z[4, 1])
    assert (tmp_path / "00000003.dat").exists() and not (tmp_path / "00000001.dat").exists()
    ek.write_eigenvectors(str(tmp_path), Z, [(5, 5)], binary=True)
    raw = (tmp_path / "00000005.dat").read_bytes()
    assert len(raw) == 248
    assert np.array_equal(np.frombuffer(raw[4:-4], dtype=np.float64), Z[:, 4])


# ----------------------------------------------------------------------------- exchange hook
def _packed_pieces(G, nb, nprow, npcol):
    """Every rank's block-cyclic piece of G, column-major, in row-major rank order."""
    out = []
    for rank in range(nprow * npcol):
        pr, pc = rank // npcol, rank % npcol
        ri = dsc.local_indices(G.shape[0], nb, pr, nprow); ci = dsc.local_indices(G.shape[1], nb, pc, npcol)
        out.append(G[np.ix_(ri, ci)].flatten(order="F"))
    return out


def virtual_allgatherv(mats, nb, nprow, npcol):
    """Exchange hook for ONE process that plays every rank in turn: call k serves the pieces of
    mats[k % len(mats)] and checks that the caller sent exactly its own piece."""
    state = {"k": 0, "rank": 0}

    def fn(send, counts, displs):
        pieces = _packed_pieces(mats[state["k"] % len(mats)], nb, nprow, npcol)
        state["k"] += 1
        assert [len(p) for p in pieces] == counts
        assert displs == [int(x) for x in np.cumsum([0] + counts[:-1])]
        assert np.array_equal(send, pieces[state["rank"]])
        return np.concatenate(pieces) if pieces else np.zeros(0)

    fn.n_ranks = nprow * npcol
    fn.state = state
    return fn


def test_gather_matrix_through_hook_single_process():
    """ek_hip_gather_matrix is pure host code: block-cyclic pieces -> full matrix on every rank."""
    rng = np.random.default_rng(11)
    lib = solver.load_library()
    try:
        for m, nb, (pr, pc) in [(30, 15, (2, 2)), (37, 8, (1, 2)), (100, 64, (2, 4)), (9, 2, (3, 1)), (5, 7, (1, 1))]:
            G = np.asfortranarray(rng.standard_normal((m, m)))
            nb = int(dsc.setup_distributed_matrix(m, m, pr, pc, 0, 0, block_size=nb)[0][dsc.BLOCK_ROW_])
            hook = virtual_allgatherv([G], nb, pr, pc)
            solver.set_allgatherv(hook)
            for rank in range(pr * pc):
                hook.state["rank"] = rank
                myrow, mycol = rank // pc, rank % pc
                desc, loc = dsc.setup_distributed_matrix(m, m, pr, pc, myrow, mycol, block_size=nb)
                assert int(desc[dsc.BLOCK_ROW_]) == nb
                ri = dsc.local_indices(m, nb, myrow, pr); ci = dsc.local_indices(m, nb, mycol, pc)
                loc[:len(ri), :len(ci)] = G[np.ix_(ri, ci)]
                proc = solver.Process(rank, pr * pc, 0, pr, pc, myrow, mycol)
                assert np.array_equal(solver.gather_matrix(loc, desc, proc), G)
        # a hook that fails, and no hook at all
        bad = lambda send, counts, displs: np.zeros(1)
        bad.n_ranks = 4
        solver.set_allgatherv(bad)
        desc, loc = dsc.setup_distributed_matrix(8, 8, 2, 2, 0, 0, block_size=2)
        with pytest.raises(solver.SolverError) as e:
            solver.gather_matrix(loc, desc, solver.Process(0, 4, 0, 2, 2, 0, 0))
        assert e.value.info == -999
        solver.set_allgatherv(None)
        with pytest.raises(solver.SolverError) as e:
            solver.gather_matrix(loc, desc, solver.Process(0, 4, 0, 2, 2, 0, 0))
        assert e.value.info == -998
        # without a hook ek_hip_solve refuses grids other than 1x1 by argument index
        ip = desc.ctypes.data_as(ctypes.POINTER(ctypes.c_int)); dp = loc.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        assert lib.ek_hip_solve(0, 8, 8, dp, ip, None, None, dp, dp, ip, 2, 2, 0, 0, None, 0) == -11
        assert lib.ek_hip_solve(0, 8, 8, dp, ip, None, None, dp, dp, ip, 1, 2, 0, 0, None, 0) == -509  # lld of a 2x2 piece
        d12, l12 = dsc.setup_distributed_matrix(8, 8, 1, 2, 0, 0, block_size=2)
        ip = d12.ctypes.data_as(ctypes.POINTER(ctypes.c_int)); dp = l12.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        assert lib.ek_hip_solve(0, 8, 8, dp, ip, None, None, dp, dp, ip, 1, 2, 0, 0, None, 0) == -12
    finally:
        solver.set_allgatherv(None)


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eigenkernel_amd import solver as sv, descriptor as d
    sv.set_allgatherv(sv.torch_allgatherv(dist))
    ok = True
    for (m, nb, pr, pc) in [(37, 8, 1, 2), (64, 16, 2, 1), (5, 1, 1, 2)]:
        G = np.asfortranarray(np.random.default_rng(5).standard_normal((m, m)))
        _, _, myrow, mycol = d.make_process_grid(rank, world, pr, pc)
        desc, loc = d.setup_distributed_matrix(m, m, pr, pc, myrow, mycol, block_size=nb)
        nbu = int(desc[d.BLOCK_ROW_])
        ri = d.local_indices(m, nbu, myrow, pr); ci = d.local_indices(m, nbu, mycol, pc)
        loc[:len(ri), :len(ci)] = G[np.ix_(ri, ci)]
        full = sv.gather_matrix(loc, desc, sv.Process(rank, world, 0, pr, pc, myrow, mycol))
        ok = ok and np.array_equal(full, G)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_matrix_gloo_world2():
    """The exchange step of ek_hip_solve for distributed inputs, two real ranks over gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_bench_promotes_the_faster_distributed_mode_that_passed():
    """bench.py --distribution auto: the replicas line stays the headline unless a distributed mode
    finished with parity ok and identical eigenvalues on every rank; then the faster such mode is
    the headline ("strong") and the replicas numbers move to "replicas"."""
    import copy
    import importlib.util
    spec = importlib.util.spec_from_file_location("ek_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    base = {"metric": "m", "value": 70000.0, "unit": "eigenpairs/s", "n_gpus": 8, "steps": 2, "warmup": 1,
            "ms_per_step": 1850.0, "scaling": "weak", "tflops_equiv": 130.0,
            "stage_seconds_per_step": {"a": 1.0}, "parity": {"p": 1}, "roofline": {"frac": 0.67},
            "config": {"workload": "N=16384 ..., full spectrum, 1 problem per GPU", "parallelism": "replicas x8"}}
    good = {"ms_per_step": 800.0, "value": 20480.0, "tflops_equiv": 38.0, "stage_seconds_per_step_rank0": {"a": 0.5},
            "parity_rank0": {"p": 2}, "roofline": {"frac": 0.3}, "parity_ok_all_ranks": True,
            "eigenvalues_identical_on_all_ranks": True}
    faster = dict(good, ms_per_step=700.0, value=23405.0)
    # nothing usable: unchanged
    out = copy.deepcopy(base)
    bench.promote_grid_mode(out, {"modes": {"collective": {"error": "x"}, "peer_windows": dict(good, parity_ok_all_ranks=False)}}, 8)
    assert out == base
    # both passed: the faster one
    out = copy.deepcopy(base)
    bench.promote_grid_mode(out, {"modes": {"collective": good, "peer_windows": faster}}, 8)
    assert out["headline_mode"] == "peer_windows" and out["scaling"] == "strong"
    assert out["value"] == 23405.0 and out["ms_per_step"] == 700.0 and out["parity"] == {"p": 2}
    assert out["replicas"]["value"] == 70000.0 and out["replicas"]["scaling"] == "weak"
    assert "ONE problem on all GPUs" in out["config"]["workload"] and "peer_windows" in out["config"]["parallelism"]
    assert out["metric"] == "m" and out["n_gpus"] == 8 and out["steps"] == 2
    # only the collective passed
    out = copy.deepcopy(base)
    bench.promote_grid_mode(out, {"modes": {"collective": good, "peer_windows": {"error": "timeout"}}}, 8)
    assert out["headline_mode"] == "collective" and out["value"] == 20480.0


def test_bench_configs_and_flop_counts_follow_baseline_and_survey():
    """bench.py --config names are BASELINE.json's configs[1..4]; flops() is SURVEY.md 8(d)'s F(N)."""
    import json
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert bench.CONFIGS["c3"] == (16384, "gep", 0) and "N=16384 generalized" in base["configs"][2]
    assert bench.CONFIGS["c2"] == (4096, "sep", 0) and "N=4096 standard" in base["configs"][1]
    assert bench.CONFIGS["c4"] == (32768, "gep", 0) and "N=32768" in base["configs"][3]
    assert bench.CONFIGS["c5"] == (16384, "gep", 1024) and "-n 1024" in base["configs"][4]
    assert abs(bench.flops(0, 4096, 4096) - 3.207e11) <= 1e8           # SEP full: 14/3 N^3
    assert abs(bench.flops(1, 16384, 16384) - 3.079e13) <= 1e10        # GEP full: 7 N^3
    assert abs(bench.flops(1, 16384, 1024) - 1.255e13) <= 1e10         # partial GEP, k = 1024
    assert bench.FP64_MFMA_PEAK_TFLOPS == 78.6 and bench.HBM_PEAK_GBS == 8000.0


def test_fortran_mpi_host_argument_contract(tmp_path):
    """host/eigenkernel_hip_mpi_app.f90 under mpiexec, the part that needs no GPU: an unknown solver ends every
    rank with "[Error] eigen_solver: Unknown solver ..." (solver_main.f90:98) and a non-zero exit code, -n is
    refused for solvers that are not *_select (command_argument.f90:186-200)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "host", "eigenkernel_hip_mpi_app")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(mpiexec) and os.path.exists("/opt/rocm/lib/llvm/bin/flang")):
        pytest.skip("needs MPICH and flang")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "host"), "eigenkernel_hip_mpi_app"])
    bad = subprocess.run([mpiexec, "-np", "2", exe, "-s", "general_elpa1", "--synthetic", "64"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "[Error] eigen_solver: Unknown solver general_elpa1" in bad.stderr
    bad = subprocess.run([mpiexec, "-np", "2", exe, "-s", "hip", "-n", "5", "--synthetic", "64"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "-n is only legal" in bad.stderr


def test_bench_reports_no_traffic_from_a_stale_pmc_record(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from a committed rocprofv3 PMC record that carries the sha256 of the kernel's
    source file: a record measured with another ek_sb2st.hip (or for another shape) must not be reported."""
    import hashlib
    import importlib
    import json
    bench = importlib.import_module("bench")
    root = tmp_path
    (root / "profiles").mkdir()
    (root / "eigenkernel_amd" / "csrc").mkdir(parents=True)
    src = root / "eigenkernel_amd" / "csrc" / "ek_sb2st.hip"
    src.write_text("// kernel source, version 1\n")
    sha = hashlib.sha256(src.read_bytes()).hexdigest()
    rec = {"n": 16384, "ncols": 16384, "hbm_bytes_per_launch": 1.0e12, "source_sha256": sha, "git": "abc1234"}
    (root / "profiles" / "r05_q2_apply_traffic.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(root))
    t, why = bench.q2_traffic_record(16384, 16384)
    assert t == 1.0e12 and "abc1234" in why
    t, why = bench.q2_traffic_record(4096, 4096)
    assert t is None and "another shape" in why
    src.write_text("// kernel source, version 2\n")
    t, why = bench.q2_traffic_record(16384, 16384)
    assert t is None and "stale" in why


def test_workspace_of_a_team_member_shrinks_with_the_team():
    """the reference allocates numroc x numroc per rank (distribute_matrix.f90:128-138).  Here the
    operators every rank applies in full to its own eigenvector columns (L, the reflectors of both stages) stay whole;
    everything else is laid out by lifetime (ek_solve.hip plan_path): the matrix and the bulge chasing's reflectors share
    one array, the D&C's bases (compact for a cell's share), Q2's records and the scratch of the other stages another,
    and the eigenvector columns are the cell's own.  Round 3 asked 9.5 padded matrices + 0.13 GB on EVERY rank whatever
    the team (DESIGN.md section 6 of that round: 20 GiB at N = 16384, 81 GiB at N = 32768).  Pure host arithmetic."""
    from eigenkernel_amd import solver
    for n in (16384, 32768):
        mat = 8 * n * n
        r3_any_p = 9.5 * mat + 0.13e9                       # round 3: every rank, every team size
        p1, parts1 = solver.workspace_bytes(1, n, n, 1)
        p2, _ = solver.workspace_bytes(1, n, n, 2)
        p4, _ = solver.workspace_bytes(1, n, n, 4)
        p8, parts8 = solver.workspace_bytes(1, n, n, 8)
        assert parts1[0] == mat
        assert p8 <= 0.55 * r3_any_p, (p8 / 2 ** 30, r3_any_p / 2 ** 30)        # VERDICT r3's mark, against round 3's P = 1
        assert p1 <= 0.66 * r3_any_p                         # one GPU: 6.2 matrices instead of 9.5
        assert p8 < p4 < p2 < p1
        assert p8 <= 5.25 * mat                              # L + Q1's reflectors + X0 + 1.63 (X1) + 1/8 (Z) + small (0.4: the
        #                                                      team SYMM's partial sums are 56 slots of n x 64)
        assert parts8[2] <= mat / 8 + 128 * 8 * n            # the eigenvector columns are the cell's share
    # a *_select arm on one GPU keeps the D&C compact too (C5: 1024 of 16384 columns)
    c5, _ = solver.workspace_bytes(1, 16384, 1024, 1)
    assert c5 <= 5.0 * 8 * 16384 * 16384


def test_bench_host_path_child_reports_instead_of_raising():
    """bench.py measures value_incl_copies in a child process that loads the library alone (the system's HIP runtime, not
    the one PyTorch bundles).  Without a GPU the child cannot initialise the library: the parent must get an error record,
    never an exception -- an optional extra does not take the bench line down."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ek_bench_child", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    out = bench.host_path_child(1, 256, 256, timeout=120.0)
    assert isinstance(out, dict)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        assert "error" in out or out.get("seconds", 0) > 0
    else:
        assert "error" in out and "child rc=" in out["error"]


def test_bench_gpus_flag_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher (no RANK in the environment) starts two ranks itself as fresh
    child processes and relays rank 0's line: n_gpus is the number of ranks that answered, not the flag echoed
    (--dry-launch: gloo, no GPU call anywhere).  Run from a directory that holds a stray module named like one
    of the standard library's: the launcher's imports must not pick it up."""
    import subprocess
    (tmp_path / "bisect.py").write_text("raise SystemExit('a stray module shadowed the standard library')\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_launch"] is True
    assert "starting 2 ranks" in out.stderr
    # under a launcher the flag does not start a second generation of ranks
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "starting" not in out.stderr and "the launcher started 1 ranks" in out.stderr
    assert json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
