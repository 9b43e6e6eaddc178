"""Host-side mirror of the reference's solver dispatch, bound to libek_hip.so via ctypes.

Mirrors src/solver_main.f90:22-100 (`eigen_solver(arg, matrix_A, eigenpairs, proc,
matrix_B)`): the `-s <name>` string selects a back-end; the new arms `hip`, `hip_select`,
`general_hip`, `general_hip_select` forward to the C-ABI (include/ek_hip.h) exactly where
the reference forwards `scalapack`, `scalapack_select`, `general_scalapack`,
`general_scalapack_select` to ScaLAPACK (:55-75).

There is NO CPU fallback here: if libek_hip.so is missing or no GPU is visible the call
raises (LibraryMissing / RuntimeError), as the reference `terminate`s when built without a
back-end (solver_elpa_dummy.f90:21).
"""
import ctypes
import os
from dataclasses import dataclass, field

import numpy as np

from . import descriptor as _d

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EK_HIP_LIB") or os.path.join(_HERE, "csrc", "libek_hip.so")   # env: A/B builds

N_STAGES = 8
SOLVERS = ("hip", "hip_select", "general_hip", "general_hip_select")
# reference solver name each arm replaces (solver_main.f90:55,59,64,66)
REPLACES = {"hip": "scalapack", "hip_select": "scalapack_select",
            "general_hip": "general_scalapack", "general_hip_select": "general_scalapack_select"}


class LibraryMissing(RuntimeError):
    pass


_lib = None
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_llp = ctypes.POINTER(ctypes.c_longlong)
# ek_hip_allgatherv_fn (include/ek_hip.h)
ALLGATHERV_FN = ctypes.CFUNCTYPE(ctypes.c_int, _dp, ctypes.c_longlong, _dp, _llp, _llp, ctypes.c_void_p)
_hook_keepalive = None


def load_library(path=None):
    """Loads libek_hip.so and declares every symbol of include/ek_hip.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise LibraryMissing(
            "libek_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C eigenkernel_amd/csrc` (there is no CPU fallback)" % p)
    lib = ctypes.CDLL(p)
    c_int, c_dbl, c_ull = ctypes.c_int, ctypes.c_double, ctypes.c_ulonglong
    vp = ctypes.c_void_p
    sigs = {
        "ek_hip_version": (c_int, []),
        "ek_hip_init": (c_int, [c_int]),
        "ek_hip_finalize": (c_int, []),
        "ek_hip_stage_name": (ctypes.c_char_p, [c_int]),
        "ek_hip_solve": (c_int, [c_int, c_int, c_int, _dp, _ip, _dp, _ip, _dp, _dp, _ip,
                                 c_int, c_int, c_int, c_int, _dp, c_int]),
        "ek_hip_solve_device": (c_int, [c_int, c_int, c_int, vp, c_int, vp, c_int, vp, vp, c_int,
                                        _dp, c_int]),
        "ek_hip_solve_replicated": (c_int, [c_int, c_int, c_int, _dp, c_int, _dp, c_int, _dp, _dp, _ip,
                                            c_int, c_int, c_int, c_int, _dp, c_int]),
        "ek_hip_solve_device_grid": (c_int, [c_int, c_int, c_int, vp, c_int, vp, c_int, vp, vp, c_int,
                                             c_int, c_int, c_int, c_int, c_int, _dp, c_int]),
        "ek_hip_set_allgatherv": (c_int, [ALLGATHERV_FN, vp]),
        "ek_hip_gather_matrix": (c_int, [c_int, c_int, _dp, _ip, c_int, c_int, c_int, c_int, _dp, c_int]),
        "ek_hip_potrf": (c_int, [c_int, _dp, _ip]),
        "ek_hip_sygst": (c_int, [c_int, _dp, _ip, _dp, _ip, _dp]),
        "ek_hip_sytrd": (c_int, [c_int, _dp, _ip, _dp, _dp, _dp]),
        "ek_hip_sytrd_team": (c_int, [c_int, _dp, _ip, _dp, _dp, _dp, c_int, _llp]),
        "ek_hip_sygst_team": (c_int, [c_int, _dp, _ip, _dp, _ip, c_int]),
        "ek_hip_potrf_team": (c_int, [c_int, _dp, _ip, c_int, _llp]),
        "ek_hip_comm_unique_id": (c_int, [vp, c_int]),
        "ek_hip_comm_init": (c_int, [vp, c_int, c_int, c_int]),
        "ek_hip_comm_attach_host": (c_int, [c_int, c_int]),
        "ek_hip_comm_size": (c_int, []),
        "ek_hip_comm_rank": (c_int, []),
        "ek_hip_comm_destroy": (c_int, []),
        "ek_hip_comm_allreduce_device": (c_int, [vp, ctypes.c_longlong]),
        "ek_hip_stedc": (c_int, [c_int, _dp, _dp, _dp, _ip]),
        "ek_hip_ormtr": (c_int, [c_int, c_int, _dp, _ip, _dp, _dp, _ip]),
        "ek_hip_trtrs": (c_int, [c_int, c_int, _dp, _ip, _dp, _ip]),
        "ek_hip_dgemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_dbl, _dp, c_int, _dp, c_int,
                                 c_dbl, _dp, c_int, c_int]),
        "ek_hip_malloc": (c_int, [ctypes.POINTER(vp), c_ull]),
        "ek_hip_free": (c_int, [vp]),
        "ek_hip_memcpy_h2d": (c_int, [vp, vp, c_ull]),
        "ek_hip_memcpy_d2h": (c_int, [vp, vp, c_ull]),
        "ek_hip_synchronize": (c_int, []),
        "ek_hip_synth_matrix_device": (c_int, [c_int, c_ull, vp, c_int]),
        "ek_hip_residual_device": (c_int, [c_int, c_int, c_int, vp, c_int, vp, c_int, vp, vp, c_int, _dp, _dp, _dp]),
        "ek_hip_orthogonality_device": (c_int, [c_int, c_int, c_int, c_int, vp, c_int, vp, c_int, _dp]),
        "ek_hip_ipratios_device": (c_int, [c_int, c_int, c_int, vp, c_int, vp, c_int, _dp]),
        "ek_hip_check": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _dp, _ip, _dp, _ip, _dp, _dp, _ip, _dp]),
        "ek_hip_profile_symv": (c_int, [c_int]),
        "ek_hip_debug_sytrd": (c_int, [c_int, c_int, c_int, _dp]),
        "ek_hip_debug_sytrd_team": (c_int, [c_int, c_int, c_int, _dp]),
        "ek_hip_debug_set_sytrd_maxcols": (c_int, [c_int]),
        "ek_hip_debug_sytrd_work_bytes": (ctypes.c_ulonglong, [c_int]),
        "ek_hip_debug_sytrd_split": (c_int, [vp, c_int]),
        "ek_hip_debug_gemm_at": (c_int, [c_int, c_int, c_int, c_int, c_int, vp, c_int, vp, c_int, c_dbl, vp, c_int,
                                         c_int, c_int, _dp]),
        "ek_hip_debug_sytrd_at": (c_int, [c_int, c_int, c_int, vp, vp, vp, _dp]),
        "ek_hip_debug_reduce_team": (c_int, [c_int, c_int, c_int, _dp]),
        "ek_hip_profile_symv_get": (c_int, [_dp, ctypes.POINTER(ctypes.c_longlong), _dp]),
        "ek_hip_debug_sy2sb": (c_int, [c_int, _dp, c_int, _dp, c_int, _dp, _ip]),
        "ek_hip_debug_sb2st": (c_int, [c_int, _dp, c_int, _dp, _dp, _dp, c_int, c_int, _ip]),
        "ek_hip_debug_two_stage_timing": (c_int, [c_int, c_int, c_int, _dp, _ip]),
        "ek_hip_debug_set_two_stage": (c_int, [c_int]),
        "ek_hip_profile_kernels": (c_int, [c_int]),
        "ek_hip_profile_kernels_get": (c_int, [_dp, _llp]),
        "ek_hip_debug_last_solve_stats": (c_int, [_dp, c_int]),
        "ek_hip_debug_sy2sb_team": (c_int, [c_int, _dp, c_int, _dp, c_int, _dp, c_int, _ip, _llp]),
        "ek_hip_debug_sy2sb_team_timing": (c_int, [c_int, c_int, c_int, _dp]),
        "ek_hip_debug_sy2sb_team_profile": (c_int, [c_int, c_int, c_int, c_int, _dp, _dp]),
        "ek_hip_debug_fail_next_chase": (c_int, [c_int]),
        "ek_hip_debug_stedc_team": (c_int, [c_int, c_int, c_int]),
        "ek_hip_debug_potrf_team_profile": (c_int, [c_int, c_int]),
        "ek_hip_debug_potrf_team_profile_get": (c_int, [c_int, _dp]),
        "ek_hip_debug_stedc_team_get": (c_int, [_dp]),
        "ek_hip_debug_last_pipe_stats": (c_int, [_dp, c_int]),
        "ek_hip_debug_workspace_bytes": (ctypes.c_ulonglong, [c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_ulonglong)]),
    }
    for name, (res, args) in sigs.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if os.environ.get("EK_HIP_BRINGUP") == "1":   # partial library during development
                continue
            raise   # header / library mismatch
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "ek_hip_version", "ek_hip_init", "ek_hip_finalize", "ek_hip_stage_name", "ek_hip_solve",
    "ek_hip_solve_device", "ek_hip_solve_replicated", "ek_hip_solve_device_grid",
    "ek_hip_set_allgatherv", "ek_hip_gather_matrix", "ek_hip_potrf", "ek_hip_sygst", "ek_hip_sytrd", "ek_hip_stedc",
    "ek_hip_ormtr", "ek_hip_trtrs", "ek_hip_dgemm", "ek_hip_malloc", "ek_hip_free",
    "ek_hip_memcpy_h2d", "ek_hip_memcpy_d2h", "ek_hip_synchronize", "ek_hip_synth_matrix_device",
    "ek_hip_profile_symv", "ek_hip_profile_symv_get", "ek_hip_debug_sytrd",
    "ek_hip_residual_device", "ek_hip_orthogonality_device", "ek_hip_ipratios_device", "ek_hip_check",
    "ek_hip_sytrd_team", "ek_hip_comm_unique_id", "ek_hip_comm_init", "ek_hip_comm_size", "ek_hip_comm_rank",
    "ek_hip_comm_destroy", "ek_hip_comm_allreduce_device", "ek_hip_debug_sytrd_team", "ek_hip_sygst_team",
    "ek_hip_potrf_team", "ek_hip_debug_reduce_team", "ek_hip_comm_attach_host",
    "ek_hip_debug_set_sytrd_maxcols", "ek_hip_debug_sytrd_work_bytes", "ek_hip_debug_sytrd_at", "ek_hip_debug_sytrd_split", "ek_hip_debug_gemm_at",
    "ek_hip_debug_sy2sb", "ek_hip_debug_sb2st", "ek_hip_debug_two_stage_timing", "ek_hip_debug_set_two_stage",
    "ek_hip_profile_kernels", "ek_hip_profile_kernels_get", "ek_hip_debug_last_solve_stats",
    "ek_hip_debug_sy2sb_team", "ek_hip_debug_sy2sb_team_timing", "ek_hip_debug_sy2sb_team_profile", "ek_hip_debug_workspace_bytes", "ek_hip_debug_fail_next_chase", "ek_hip_debug_last_pipe_stats",
    "ek_hip_debug_stedc_team", "ek_hip_debug_stedc_team_get",
    "ek_hip_debug_potrf_team_profile", "ek_hip_debug_potrf_team_profile_get",
)


def _check(what, A, B, w, Z, n_cols, index1=1, index2=1):
    lib = load_library()
    Z = _farr(Z)
    n = Z.shape[0]
    A_ = _farr(A) if A is not None else None
    B_ = _farr(B) if B is not None else None
    out = np.zeros(max(3, n))
    wv = np.ascontiguousarray(np.asarray(w, dtype=np.float64)) if w is not None else np.zeros(max(n, 1))
    info = lib.ek_hip_check(what, 1 if B is not None else 0, n, n_cols, index1, index2,
                            _P(A_) if A_ is not None else None, _I(_desc_for(A_)) if A_ is not None else None,
                            _P(B_) if B_ is not None else None, _I(_desc_for(B_)) if B_ is not None else None,
                            _P(wv), _P(Z), _I(_desc_for(Z)), _P(out))
    if info:
        raise RuntimeError("ek_hip_check(%d) info=%d" % (what, info))
    return out


def eval_residual_norm(A, values, V, B=None, n_check=None):
    """verifier.f90:207 on the GPU. Returns (A_norm, res_norm_ave, res_norm_max)."""
    n_check = V.shape[1] if n_check is None else n_check
    Zf = np.zeros((V.shape[0], V.shape[0]), order="F"); Zf[:, :V.shape[1]] = V
    wv = np.zeros(V.shape[0]); wv[:len(values)] = values
    out = _check(0, A, B, wv, Zf, n_check)
    return out[0], out[1], out[2]


def eval_orthogonality(V, B=None, index1=1, index2=None):
    """verifier.f90:333 on the GPU."""
    index2 = V.shape[1] if index2 is None else index2
    Zf = np.zeros((V.shape[0], V.shape[0]), order="F"); Zf[:, :V.shape[1]] = V
    return _check(1, None, B, None, Zf, 0, index1, index2)[0]


def get_ipratios(V, S=None, n_vec=None):
    """distribute_matrix.f90:18 on the GPU."""
    n_vec = V.shape[1] if n_vec is None else n_vec
    Zf = np.zeros((V.shape[0], V.shape[0]), order="F"); Zf[:, :V.shape[1]] = V
    return _check(2, None, S, None, Zf, n_vec)[:n_vec].copy()



def _P(a):
    return a.ctypes.data_as(_dp)


def _I(a):
    return a.ctypes.data_as(_ip)


def _farr(a):
    a = np.asarray(a, dtype=np.float64)
    return a if a.flags.f_contiguous else np.asfortranarray(a)


def _desc_for(a, nb=None):
    m, n = a.shape
    nb = min(_d.g_block_size if nb is None else nb, max(min(m, n), 1))
    return _d.descinit(m, n, nb, nb, 0, 0, 0, max(1, a.strides[1] // 8 if n > 1 else m))


# ----------------------------------------------------------------------------- exchange hook
def set_allgatherv(fn):
    """Registers the host exchange hook ek_hip_solve needs on grids larger than 1x1.

    fn(send: ndarray[count], counts: list[int], displs: list[int]) -> ndarray[sum(counts)]
    with MPI_Allgatherv semantics over the grid's ranks in row-major order; None removes it.
    """
    global _hook_keepalive
    lib = load_library()
    if fn is None:
        lib.ek_hip_set_allgatherv(ctypes.cast(None, ALLGATHERV_FN), None)
        _hook_keepalive = None
        return

    n_ranks = getattr(fn, "n_ranks", None)
    if n_ranks is None:
        raise ValueError("the hook must carry .n_ranks (number of ranks of the grid)")

    def thunk(send, count, recv, counts, displs, _user):
        try:
            cs = [int(counts[r]) for r in range(n_ranks)]
            ds = [int(displs[r]) for r in range(n_ranks)]
            count = int(count)
            sv = np.ctypeslib.as_array(send, shape=(max(count, 1),))[:count]
            out = np.asarray(fn(sv, cs, ds), dtype=np.float64)
            total = ds[-1] + cs[-1]
            if out.shape != (total,):
                return 2
            np.ctypeslib.as_array(recv, shape=(max(total, 1),))[:total] = out
            return 0
        except Exception:                     # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1

    cb = ALLGATHERV_FN(thunk)
    _hook_keepalive = (cb, thunk, fn)
    lib.ek_hip_set_allgatherv(cb, None)


def torch_allgatherv(dist, group=None):
    """Exchange hook on torch.distributed host tensors (gloo in the CPU tests; with the nccl
    backend pass a gloo side group): pads every piece to the largest and all-gathers."""
    import torch
    world = dist.get_world_size(group)

    def fn(send, counts, displs):
        mx = max(max(counts), 1)
        mine = torch.zeros(mx, dtype=torch.float64)
        mine[:send.shape[0]] = torch.from_numpy(np.ascontiguousarray(send))
        parts = [torch.empty(mx, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        out = np.empty(displs[-1] + counts[-1])
        for r in range(world):
            out[displs[r]:displs[r] + counts[r]] = parts[r][:counts[r]].numpy()
        return out

    fn.n_ranks = world
    return fn


def gather_matrix(M_loc, desc, proc):
    """ek_hip_gather_matrix: the full matrix from every rank's block-cyclic piece (host only)."""
    lib = load_library()
    m, n = int(desc[_d.ROWS_]), int(desc[_d.COLS_])
    full = np.zeros((m, n), order="F")
    info = lib.ek_hip_gather_matrix(m, n, _P(M_loc), _I(desc), proc.n_procs_row, proc.n_procs_col,
                                    proc.my_proc_row, proc.my_proc_col, _P(full), max(1, m))
    if info != 0:
        raise SolverError("ek_hip_gather_matrix failed", info)
    return full


# ----------------------------------------------------------------------------- types
@dataclass
class Process:
    """ek_process_t (processes.f90:6-9)."""
    my_rank: int = 0
    n_procs: int = 1
    context: int = 0
    n_procs_row: int = 1
    n_procs_col: int = 1
    my_proc_row: int = 0
    my_proc_col: int = 0


@dataclass
class EigenpairsBlacs:
    """ek_eigenpairs_blacs_t (eigenpairs_types.f90:7-11); type_number = 2."""
    values: np.ndarray = None
    desc: np.ndarray = None
    Vectors: np.ndarray = None
    type_number: int = 2
    stage_seconds: dict = field(default_factory=dict)
    info: int = 0


class SolverError(RuntimeError):
    """Raised where the reference calls terminate(msg, info) (processes.f90:122-139)."""

    def __init__(self, msg, info):
        super().__init__("%s (info=%d)" % (msg, info))
        self.info = info


# ----------------------------------------------------------------------------- stage-level wrappers
def potrf(B):
    """PDPOTRF('L') (generalized_to_standard.f90:24). Returns (B_with_L_in_lower, info)."""
    lib = load_library()
    B = np.array(_farr(B), order="F", copy=True)
    desc = _desc_for(B)
    info = lib.ek_hip_potrf(B.shape[0], _P(B), _I(desc))
    return B, info


def sygst(A, L):
    """PDSYGST(1,'L') (generalized_to_standard.f90:37)."""
    lib = load_library()
    A = np.array(_farr(A), order="F", copy=True)
    L = _farr(L)
    scale = ctypes.c_double(0.0)
    info = lib.ek_hip_sygst(A.shape[0], _P(A), _I(_desc_for(A)), _P(L), _I(_desc_for(L)),
                            ctypes.byref(scale))
    return A, info


def sytrd(A):
    """PDSYTRD('L') (solver_scalapack_all.f90:59). Returns (A_reflectors, d, e, tau, info)."""
    lib = load_library()
    A = np.array(_farr(A), order="F", copy=True)
    n = A.shape[0]
    d = np.zeros(max(n, 1)); e = np.zeros(max(n, 1)); tau = np.zeros(max(n, 1))
    info = lib.ek_hip_sytrd(n, _P(A), _I(_desc_for(A)), _P(d), _P(e), _P(tau))
    return A, d[:n], e[:max(n - 1, 0)], tau[:max(n - 1, 0)], info


def sytrd_team(A, nteam):
    """PDSYTRD('L') on a 1 x P grid, 128-wide column blocks (ek_hip_sytrd_team).  nteam >= 1: the
    whole team rehearsed in this process on one GPU; nteam == 0: this process is one rank of the
    attached communicator.  Returns (A_reflectors, d, e, tau, info, mismatch)."""
    lib = load_library()
    A = np.array(_farr(A), order="F", copy=True)
    n = A.shape[0]
    d = np.zeros(max(n, 1)); e = np.zeros(max(n, 1)); tau = np.zeros(max(n, 1))
    mm = ctypes.c_longlong(-1)
    info = lib.ek_hip_sytrd_team(n, _P(A), _I(_desc_for(A)), _P(d), _P(e), _P(tau), nteam, ctypes.byref(mm))
    return A, d[:n], e[:max(n - 1, 0)], tau[:max(n - 1, 0)], info, mm.value


def potrf_team(B, nteam):
    """PDPOTRF('L') on a 1 x P grid (ek_hip_potrf_team). Returns (B_with_L_in_lower, info, mismatch)."""
    lib = load_library()
    B = np.array(_farr(B), order="F", copy=True)
    mm = ctypes.c_longlong(-1)
    info = lib.ek_hip_potrf_team(B.shape[0], _P(B), _I(_desc_for(B)), nteam, ctypes.byref(mm))
    return B, info, mm.value


def sygst_team(A, L, nteam):
    """PDSYGST(1,'L') on a 1 x P grid (ek_hip_sygst_team). Returns (reduced A, info)."""
    lib = load_library()
    A = np.array(_farr(A), order="F", copy=True)
    L = _farr(L)
    n = A.shape[0]
    info = lib.ek_hip_sygst_team(n, _P(A), _I(_desc_for(A)), _P(L), _I(_desc_for(L)), nteam)
    return A, info


def comm_unique_id():
    """128-byte RCCL id (rank 0 calls this, the host broadcasts it)."""
    lib = load_library()
    buf = ctypes.create_string_buffer(128)
    rc = lib.ek_hip_comm_unique_id(buf, 128)
    if rc:
        raise SolverError("ek_hip_comm_unique_id: %d" % rc, rc)
    return buf.raw


def comm_init(uid, nranks, rank):
    """Attach the RCCL communicator of the distributed path (include/ek_hip.h)."""
    lib = load_library()
    buf = ctypes.create_string_buffer(bytes(uid), 128)
    rc = lib.ek_hip_comm_init(buf, 128, nranks, rank)
    if rc:
        raise SolverError("ek_hip_comm_init: %d" % rc, rc)


def comm_attach_host(nranks, rank):
    """Distributed stages with the exchanges going through the registered allgatherv hook."""
    rc = load_library().ek_hip_comm_attach_host(nranks, rank)
    if rc:
        raise SolverError("ek_hip_comm_attach_host: %d" % rc, rc)


def comm_destroy():
    load_library().ek_hip_comm_destroy()


def stedc(d, e):
    """PDSTEDC('I') (solver_scalapack_all.f90:96). Returns (w, Z, info)."""
    lib = load_library()
    d = np.array(d, dtype=np.float64, copy=True)
    n = d.shape[0]
    ee = np.zeros(max(n, 1)); ee[:max(n - 1, 0)] = np.asarray(e, dtype=np.float64)[:max(n - 1, 0)]
    Z = np.zeros((n, n), order="F")
    info = lib.ek_hip_stedc(n, _P(d), _P(ee), _P(Z), _I(_desc_for(Z)))
    return d, Z, info


def ormtr(Ar, tau, Z):
    """PDORMTR('L','L','N') (solver_scalapack_all.f90:115). Returns (QZ, info)."""
    lib = load_library()
    Ar = _farr(Ar)
    Z = np.array(_farr(Z), order="F", copy=True)
    n = Ar.shape[0]
    t = np.zeros(max(n, 1)); t[:max(n - 1, 0)] = np.asarray(tau, dtype=np.float64)[:max(n - 1, 0)]
    info = lib.ek_hip_ormtr(n, Z.shape[1], _P(Ar), _I(_desc_for(Ar)), _P(t), _P(Z), _I(_desc_for(Z)))
    return Z, info


def trtrs(L, Z):
    """PDTRTRS('L','T','N') (generalized_to_standard.f90:103). Returns (X, info)."""
    lib = load_library()
    L = _farr(L)
    Z = np.array(_farr(Z), order="F", copy=True)
    info = lib.ek_hip_trtrs(L.shape[0], Z.shape[1], _P(L), _I(_desc_for(L)), _P(Z), _I(_desc_for(Z)))
    return Z, info


BAND_W = 64   # half bandwidth of the two-stage tridiagonalisation (kBandW in csrc/ek_common.h)


def set_two_stage(min_order=-1):
    """Order from which the whole-path calls tridiagonalise in two stages (-1: default, 0: never)."""
    load_library().ek_hip_debug_set_two_stage(int(min_order))


def stedc_team(nranks=0, levels=-1, profile=False):
    """Team form of the divide & conquer's heights below the top merge (ek_stedc.hip StedcTeam).  levels: sharded heights
    (-1: the library's default for the order and team) -- applies to real teams too; nranks >= 2: grid cells solved
    WITHOUT a communicator rehearse a team of that many, rank after rank in this process.  stedc_team() restores the
    defaults."""
    rc = load_library().ek_hip_debug_stedc_team(int(nranks), int(levels), 1 if profile else 0)
    assert rc == 0, rc


def stedc_team_seconds():
    """[the D&C, all ranks' sections, the longest rank's section per height summed] of the last profiled rehearsal."""
    out = (ctypes.c_double * 3)()
    rc = load_library().ek_hip_debug_stedc_team_get(out)
    assert rc == 0, rc
    return [float(x) for x in out]


def last_solve_stats():
    """What the last whole-path call did: [flops of the D&C's merge products, 1 if it stayed on the two-stage path,
    panels of the first stage that took the Householder rescue, 1 if the band short cut was taken, ...]."""
    st = (ctypes.c_double * 8)()
    load_library().ek_hip_debug_last_solve_stats(st, 8)
    return [float(x) for x in st]


def workspace_bytes(problem, n, n_vec=None, nranks=1):
    """Bytes of device workspace one whole-path call asks for (host arithmetic: works without a GPU): returns
    (total, parts) with parts = [one matrix, L + Q1 reflectors, eigenvector columns, X0, X1, rest]."""
    parts = (ctypes.c_ulonglong * 6)()
    tot = load_library().ek_hip_debug_workspace_bytes(int(problem), int(n), int(n if n_vec is None else n_vec), int(nranks), parts)
    return int(tot), [int(x) for x in parts]


def sy2sb(A):
    """Stage 1 of the two-stage tridiagonalisation on its own (include/ek_hip_debug.h):
    returns (A_out with the band in its lower band, V explicit reflectors, tau, flag)."""
    lib = load_library()
    A = np.array(_farr(A), order="F", copy=True)
    n = A.shape[0]
    V = np.zeros((n, n), order="F"); tau = np.zeros(max(n, 1)); flag = ctypes.c_int(-1)
    rc = lib.ek_hip_debug_sy2sb(n, _P(A), max(1, n), _P(V), max(1, n), _P(tau), ctypes.byref(flag))
    if rc:
        raise RuntimeError("ek_hip_debug_sy2sb info=%d" % rc)
    return A, V, tau[:n], flag.value


def sy2sb_team(A, nteam):
    """Stage 1 over a team (rehearsed inside this process for nteam >= 1; nteam = 0: one rank of the attached
    communicator): returns (band in the lower band of an otherwise zero matrix, V, tau, flag, mismatch)."""
    lib = load_library()
    A = np.array(_farr(A), order="F", copy=True)
    n = A.shape[0]
    V = np.zeros((n, n), order="F"); tau = np.zeros(max(n, 1)); flag = ctypes.c_int(-1); mism = ctypes.c_longlong(-1)
    rc = lib.ek_hip_debug_sy2sb_team(n, _P(A), max(1, n), _P(V), max(1, n), _P(tau), int(nteam), ctypes.byref(flag),
                                     ctypes.byref(mism))
    if rc:
        raise RuntimeError("ek_hip_debug_sy2sb_team info=%d" % rc)
    return A, V, tau[:n], flag.value, mism.value


def sb2st(Bd, Z=None):
    """Stage 2 on its own: the lower band (half bandwidth 64) of Bd -> (d, e, Q2 Z or None, flag)."""
    lib = load_library()
    Bd = np.array(_farr(Bd), order="F", copy=True)
    n = Bd.shape[0]
    d = np.zeros(max(n, 1)); e = np.zeros(max(n, 1)); flag = ctypes.c_int(-1)
    if Z is None:
        rc = lib.ek_hip_debug_sb2st(n, _P(Bd), max(1, n), _P(d), _P(e), None, max(1, n), 0, ctypes.byref(flag))
    else:
        Z = np.array(_farr(Z), order="F", copy=True)
        rc = lib.ek_hip_debug_sb2st(n, _P(Bd), max(1, n), _P(d), _P(e), _P(Z), max(1, n), Z.shape[1], ctypes.byref(flag))
    if rc:
        raise RuntimeError("ek_hip_debug_sb2st info=%d" % rc)
    return d[:n], e[:max(n - 1, 0)], Z, flag.value


def dgemm(transa, transb, alpha, A, B, beta, C, lower_only=False):
    lib = load_library()
    A = _farr(A); B = _farr(B)
    C = np.array(_farr(C), order="F", copy=True)
    m, n = C.shape
    k = A.shape[0] if transa else A.shape[1]
    info = lib.ek_hip_dgemm(int(transa), int(transb), m, n, k, alpha, _P(A), max(1, A.shape[0]),
                            _P(B), max(1, B.shape[0]), beta, _P(C), max(1, m), int(lower_only))
    if info:
        raise RuntimeError("ek_hip_dgemm info=%d" % info)
    return C


# ----------------------------------------------------------------------------- the dispatch
def eigen_solver(solver_type, matrix_A, matrix_B=None, n_vec=None, block_size=None, proc=None,
                 inputs="replicated"):
    """eigen_solver (solver_main.f90:22-100) for the hip arms.

    matrix_A / matrix_B: SparseMat (replicated triplets, as the reference passes) or dense
    symmetric ndarrays.  Returns (eigenpairs: EigenpairsBlacs, proc: Process).
    With a process grid larger than 1x1 (proc.n_procs_row x proc.n_procs_col, one rank per GPU)
    every rank passes the same replicated matrices (main.f90:84-86) and receives its
    block-cyclic piece of the eigenvectors (ek_hip_solve_replicated): no collective is issued.
    inputs="distributed" reproduces the reference's own data flow instead: A and B are cut into
    block-cyclic pieces (setup_distributed_matrix + distribute_global_sparse_matrix) and handed
    to ek_hip_solve, which reassembles them through the hook of set_allgatherv().
    Raises SolverError where the reference terminates (info != 0 from the Cholesky,
    reduction or recovery stage), ValueError for an unknown solver
    ('eigen_solver: Unknown solver', solver_main.f90:98).
    """
    if solver_type not in SOLVERS:
        raise ValueError("eigen_solver: Unknown solver %r" % (solver_type,))
    generalized = solver_type.startswith("general_")
    select = solver_type.endswith("_select")
    if generalized and matrix_B is None:
        raise ValueError("eigen_solver: matrix_B is required for %s" % solver_type)
    lib = load_library()
    proc = proc or Process()
    gridded = proc.n_procs_row * proc.n_procs_col != 1

    def dense(m):
        return m.to_dense() if hasattr(m, "to_dense") else np.array(_farr(m), order="F", copy=True)

    A = dense(matrix_A)
    n = A.shape[0]
    if n_vec is None or not select:
        if n_vec is not None and n_vec != n and not select:
            raise ValueError("-n is only legal for *_select solvers (command_argument.f90:186-200)")
        n_vec = n if not select or n_vec is None else n_vec
    if gridded and inputs == "distributed":
        g = (proc.n_procs_row, proc.n_procs_col, proc.my_proc_row, proc.my_proc_col)

        def piece(M):
            desc, loc = _d.setup_distributed_matrix(n, n, *g, block_size=block_size, ctxt=proc.context)
            nb = int(desc[_d.BLOCK_ROW_])
            ri = _d.local_indices(n, nb, g[2], g[0]); ci = _d.local_indices(n, nb, g[3], g[1])
            if len(ri) and len(ci):
                loc[:len(ri), :len(ci)] = M[np.ix_(ri, ci)]
            return desc, loc

        desc_A, A_loc = piece(A)
        desc_B, B_loc = piece(dense(matrix_B)) if generalized else (None, None)
        desc_Z, Z_loc = _d.setup_distributed_matrix(n, n, *g, block_size=int(desc_A[_d.BLOCK_ROW_]),
                                                    ctxt=proc.context)
        w = np.zeros(n)
        stage = np.zeros(N_STAGES)
        info = lib.ek_hip_solve(1 if generalized else 0, n, n_vec, _P(A_loc), _I(desc_A),
                                _P(B_loc) if generalized else None,
                                _I(desc_B) if generalized else None, _P(w), _P(Z_loc), _I(desc_Z),
                                *g, _P(stage), N_STAGES)
        if info != 0:
            raise SolverError("eigen_solver(%s): libek_hip failed" % solver_type, info)
        ep = EigenpairsBlacs(values=w, desc=desc_Z, Vectors=Z_loc, info=info)
        ep.stage_seconds = {lib.ek_hip_stage_name(i).decode(): float(stage[i]) for i in range(N_STAGES)}
        ep.n_vec = n_vec
        ep.A_loc, ep.B_loc = A_loc, B_loc
        return ep, proc
    if gridded:
        B = dense(matrix_B) if generalized else None
        desc_Z, Z_loc = _d.setup_distributed_matrix(
            n, n, proc.n_procs_row, proc.n_procs_col, proc.my_proc_row, proc.my_proc_col,
            block_size=block_size, ctxt=proc.context)
        w = np.zeros(n)
        stage = np.zeros(N_STAGES)
        info = lib.ek_hip_solve_replicated(
            1 if generalized else 0, n, n_vec, _P(A), max(1, n), _P(B) if generalized else None,
            max(1, n), _P(w), _P(Z_loc), _I(desc_Z), proc.n_procs_row, proc.n_procs_col,
            proc.my_proc_row, proc.my_proc_col, _P(stage), N_STAGES)
        if info != 0:
            raise SolverError("eigen_solver(%s): libek_hip failed" % solver_type, info)
        ep = EigenpairsBlacs(values=w, desc=desc_Z, Vectors=Z_loc, info=info)
        ep.stage_seconds = {lib.ek_hip_stage_name(i).decode(): float(stage[i]) for i in range(N_STAGES)}
        ep.n_vec = n_vec
        return ep, proc
    # setup_distributed_matrix (distribute_matrix.f90:92-148) on the 1x1 grid
    desc_A, A_loc = _d.setup_distributed_matrix(n, n, block_size=block_size)
    A_loc[:, :] = A
    if generalized:
        desc_B, B_loc = _d.setup_distributed_matrix(n, n, block_size=block_size)
        B_loc[:, :] = dense(matrix_B)
    else:
        desc_B, B_loc = None, None
    desc_Z, Z_loc = _d.setup_distributed_matrix(n, n, block_size=int(desc_A[_d.BLOCK_ROW_]))
    w = np.zeros(n)
    stage = np.zeros(N_STAGES)
    info = lib.ek_hip_solve(1 if generalized else 0, n, n_vec, _P(A_loc), _I(desc_A),
                            _P(B_loc) if generalized else None,
                            _I(desc_B) if generalized else None, _P(w), _P(Z_loc), _I(desc_Z),
                            1, 1, 0, 0, _P(stage), N_STAGES)
    if info != 0:
        raise SolverError("eigen_solver(%s): libek_hip failed" % solver_type, info)
    ep = EigenpairsBlacs(values=w, desc=desc_Z, Vectors=Z_loc, info=info)
    ep.stage_seconds = {lib.ek_hip_stage_name(i).decode(): float(stage[i]) for i in range(N_STAGES)}
    ep.n_vec = n_vec
    return ep, proc
