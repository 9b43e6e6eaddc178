"""ctypes loader for oracle/libek_oracle.so (built by `make -C oracle`).

TEST INFRASTRUCTURE ONLY: the product package (eigenkernel_amd/) never imports this.
Each wrapper cites the reference call site its C function restates (see ek_oracle.c).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = ctypes.POINTER(ctypes.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libek_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    assert a.dtype == np.float64 and a.flags.f_contiguous
    return a.ctypes.data_as(_dp)


def _f(a):
    return np.asfortranarray(np.array(a, dtype=np.float64, copy=True))


def synth_matrix(n, seed):
    """SURVEY.md 8(d) generator: A = seed 1, B = seed 2."""
    A = np.zeros((n, n), order="F")
    lib().ok_synth_matrix(ctypes.c_int(n), ctypes.c_uint64(seed), _p(A), ctypes.c_int(n))
    return A


def potrf_lower(B):
    """generalized_to_standard.f90:24 PDPOTRF('L'). Returns (L_in_lower, info)."""
    B = _f(B)
    n = B.shape[0]
    info = lib().ok_potrf_lower(n, _p(B), n)
    return B, info


def sygst_lower(A, L):
    """generalized_to_standard.f90:37 PDSYGST(1,'L')."""
    A = _f(A)
    L = _f(L)
    n = A.shape[0]
    lib().ok_sygst_lower(n, _p(A), n, _p(L), n)
    return A


def sytrd_lower(A):
    """solver_scalapack_all.f90:59 PDSYTRD('L'). Returns (A_with_reflectors, d, e, tau)."""
    A = _f(A)
    n = A.shape[0]
    d = np.zeros(n)
    e = np.zeros(max(n - 1, 1))
    tau = np.zeros(max(n - 1, 1))
    lib().ok_sytrd_lower(n, _p(A), n, _p(d), _p(e), _p(tau))
    return A, d, e[: n - 1], tau[: n - 1]


def stedc(d, e, smlsiz=25):
    """solver_scalapack_all.f90:96 PDSTEDC('I'). Returns (w, Z)."""
    d = np.array(d, dtype=np.float64)
    n = d.shape[0]
    e = np.concatenate([np.array(e, dtype=np.float64), [0.0]])
    Z = np.zeros((n, n), order="F")
    info = lib().ok_stedc(n, _p(d), _p(e), _p(Z), n, smlsiz)
    assert info == 0
    return d, Z


def steqr(d, e):
    d = np.array(d, dtype=np.float64)
    n = d.shape[0]
    e = np.concatenate([np.array(e, dtype=np.float64), [0.0]])
    Z = np.zeros((n, n), order="F")
    info = lib().ok_steqr(n, _p(d), _p(e), _p(Z), n, 1)
    assert info == 0
    return d, Z


def stebz_stein(d, e, n_vec):
    """solver_scalapack_select.f90:56 PDSYEVX tridiagonal part. Returns (w[:n_vec], Z[:, :n_vec])."""
    d = np.array(d, dtype=np.float64)
    n = d.shape[0]
    e = np.concatenate([np.array(e, dtype=np.float64), [0.0]])
    w = np.zeros(n)
    Z = np.zeros((n, n), order="F")
    info = lib().ok_stebz_stein(n, n_vec, _p(d), _p(e), _p(w), _p(Z), n)
    assert info == 0
    return w[:n_vec], Z[:, :n_vec]


def ormtr_lower(Ar, tau, Z):
    """solver_scalapack_all.f90:115 PDORMTR('L','L','N')."""
    Ar = _f(Ar)
    Z = _f(Z)
    n = Ar.shape[0]
    t = np.concatenate([np.array(tau, dtype=np.float64), [0.0]])
    lib().ok_ormtr_lower(n, Z.shape[1], _p(Ar), n, _p(t), _p(Z), n)
    return Z


def trtrs_lt(L, Z):
    """generalized_to_standard.f90:103 PDTRTRS('L','T','N'). Returns (X, info)."""
    L = _f(L)
    Z = _f(Z)
    n = L.shape[0]
    info = lib().ok_trtrs_lt(n, Z.shape[1], _p(L), n, _p(Z), n)
    return Z, info


def solve(A, B=None, n_vec=None, tri_solver=0):
    """Whole path (solver_scalapack_all.f90:127-168 / :19-124). Returns (w, Z, info, L)."""
    A = _f(A)
    n = A.shape[0]
    n_vec = n if n_vec is None else n_vec
    problem = 0 if B is None else 1
    Bf = _f(B) if B is not None else np.zeros((1, 1), order="F")
    w = np.zeros(n)
    Z = np.zeros((n, n), order="F")
    info = lib().ok_solve(problem, n, n_vec, _p(A), n, _p(Bf), max(n if problem else 1, 1),
                          _p(w), _p(Z), n, tri_solver)
    return w[:n_vec], Z[:, :n_vec], info, (Bf if problem else None)
