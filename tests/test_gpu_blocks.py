"""GPU parity of the building blocks (GEMM, Cholesky, reduction, recovery) through the C-ABI.

Each HIP stage is compared with the CPU oracle (oracle/ek_oracle.c) on the same seeded
inputs, and with oracle-independent identities (||L L^T - B||, ...).  fp64 tolerances are
written next to each assertion as multiples of N * eps * scale.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EPS = 2.220446049250313e-16


def _rand(m, n, seed):
    return np.asfortranarray(np.random.default_rng(seed).uniform(-1, 1, (m, n)))


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(128, 128, 16), (64, 48, 4), (257, 130, 77), (1, 1, 1),
                                   (300, 5, 129), (513, 384, 200)])
def test_dgemm_matches_numpy(hip, ta, tb, m, n, k):
    A = _rand(k, m, 1) if ta else _rand(m, k, 1)
    B = _rand(n, k, 2) if tb else _rand(k, n, 2)
    C = _rand(m, n, 3)
    got = hip.dgemm(ta, tb, 0.75, A, B, -0.5, C)
    ref = 0.75 * (A.T if ta else A) @ (B.T if tb else B) - 0.5 * C
    assert np.abs(got - ref).max() <= 4 * k * EPS   # |a|,|b| <= 1


def test_dgemm_asymmetric_identity(hip):
    """A = I with an asymmetric B catches a row/column swap in the MFMA C-layout."""
    n = 128
    B = np.asfortranarray(np.arange(n * n, dtype=np.float64).reshape(n, n))
    got = hip.dgemm(0, 0, 1.0, np.eye(n), B, 0.0, np.zeros((n, n)))
    assert np.array_equal(got, B)


def test_dgemm_beta_zero_ignores_nan(hip):
    A, B = _rand(70, 9, 1), _rand(9, 33, 2)
    C = np.full((70, 33), np.nan)
    got = hip.dgemm(0, 0, 1.0, A, B, 0.0, C)
    assert np.abs(got - A @ B).max() <= 40 * EPS


def test_dgemm_lower_only(hip):
    n, k = 300, 40
    A = _rand(n, k, 5)
    C = _rand(n, n, 6)
    got = hip.dgemm(0, 1, -1.0, A, A, 1.0, C, lower_only=True)
    ref = C - A @ A.T
    il = np.tril_indices(n)
    assert np.abs(got[il] - ref[il]).max() <= 4 * k * EPS
    # tiles strictly above the diagonal are untouched
    assert np.array_equal(got[:128, 256:], C[:128, 256:])


@pytest.mark.parametrize("n", [1, 5, 64, 128, 129, 300, 640, 1000, 1024, 1500, 2177])   # >= 1024: look-ahead form
def test_potrf_matches_oracle(hip, oracle, n):
    B = oracle.synth_matrix(n, 2)
    L_or, info_or = oracle.potrf_lower(B)
    got, info = hip.potrf(B)
    assert info == 0 and info_or == 0
    L = np.tril(got)
    assert np.abs(L - np.tril(L_or)).max() <= 8 * n * EPS * np.abs(L_or).max()
    assert np.abs(L @ L.T - B).max() <= 8 * n * EPS * np.abs(B).max()


def test_potrf_lookahead_reports_first_bad_pivot(hip, oracle):
    n = 1300
    B = oracle.synth_matrix(n, 2)
    B[700, 700] = -1.0
    B[1100, 1100] = -1.0
    _, info = hip.potrf(B)
    assert info == 701


@pytest.mark.parametrize("bad", [150, 100, 63, 64, 0, 199])   # both 64-blocks of a 128x128 diagonal block, its edges
def test_potrf_reports_first_bad_pivot(hip, oracle, bad):
    n = 200
    B = oracle.synth_matrix(n, 2)
    B[bad, bad] = -1.0
    _, info_or = oracle.potrf_lower(B)
    _, info = hip.potrf(B)
    assert info == info_or == bad + 1


@pytest.mark.parametrize("n", [3, 100, 257, 600])
def test_sygst_matches_oracle(hip, oracle, n):
    A = oracle.synth_matrix(n, 1)
    B = oracle.synth_matrix(n, 2)
    L, _ = oracle.potrf_lower(B)
    C_or = oracle.sygst_lower(A, L)
    got, info = hip.sygst(A, L)
    assert info == 0
    il = np.tril_indices(n)
    assert np.abs(got[il] - C_or[il]).max() <= 32 * n * EPS * np.abs(C_or).max()
    Lt = np.tril(L)
    assert np.abs(Lt @ np.tril(got) @ Lt.T - 0).shape == (n, n)
    Cfull = np.tril(got) + np.tril(got, -1).T
    assert np.abs(Lt @ Cfull @ Lt.T - A).max() <= 64 * n * EPS * np.abs(A).max()


@pytest.mark.parametrize("n,nrhs", [(4, 4), (130, 130), (300, 7), (515, 515)])
def test_trtrs_matches_oracle(hip, oracle, n, nrhs):
    B = oracle.synth_matrix(n, 2)
    L, _ = oracle.potrf_lower(B)
    Z = _rand(n, nrhs, 9)
    X_or, info_or = oracle.trtrs_lt(L, Z)
    X, info = hip.trtrs(L, Z)
    assert info == 0 and info_or == 0
    assert np.abs(X - X_or).max() <= 16 * n * EPS * np.abs(X_or).max()


def test_trtrs_singular_diag(hip, oracle):
    n = 50
    L = np.tril(oracle.synth_matrix(n, 2))
    L[20, 20] = 0.0
    _, info_or = oracle.trtrs_lt(L, np.ones((n, 2)))
    _, info = hip.trtrs(L, np.ones((n, 2)))
    assert info == info_or == 21


def _tridiag_check(A, Ar, d, e, tau, oracle):
    """Q T Q^T == A with Q rebuilt from the reflectors by the oracle's ormtr."""
    n = A.shape[0]
    T = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
    Q = oracle.ormtr_lower(Ar, tau, np.eye(n))
    return np.abs(Q @ T @ Q.T - A).max(), np.abs(Q.T @ Q - np.eye(n)).max()


@pytest.mark.parametrize("n", [2, 3, 17, 64, 65, 128, 130, 257, 400, 700])
def test_sytrd_matches_oracle(hip, oracle, n):
    A = oracle.synth_matrix(n, 1)
    Ar_or, d_or, e_or, tau_or = oracle.sytrd_lower(A)
    Ar, d, e, tau, info = hip.sytrd(A)
    assert info == 0
    scale = np.abs(A).max()
    # Same Householder convention => d, e, tau, v agree element-wise, but only up to the
    # conditioning of the individual entries (the trailing reflectors amplify rounding):
    # a loose element-wise sanity bound, and the tight bounds on what IS well conditioned --
    # the backward error ||Q T Q^T - A||, the orthogonality of Q and the spectrum of T.
    assert np.abs(d - d_or).max() <= 1e-9 * scale
    assert np.abs(np.abs(e) - np.abs(e_or)).max() <= 1e-9 * scale
    assert np.abs(tau - tau_or).max() <= 1e-9
    res, orth = _tridiag_check(A, Ar, d, e, tau, oracle)
    assert res <= 64 * n * EPS * scale
    assert orth <= 64 * n * EPS
    T = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
    T_or = np.diag(d_or) + np.diag(e_or, -1) + np.diag(e_or, 1)
    assert np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(T_or)).max() <= 8 * n * EPS * scale
    assert np.abs(np.linalg.eigvalsh(T) - np.linalg.eigvalsh(A)).max() <= 8 * n * EPS * scale
    il = np.tril_indices(n, -2)
    if len(il[0]):
        assert np.abs(Ar[il] - Ar_or[il]).max() <= 1e-8
