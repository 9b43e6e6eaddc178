"""Host-side mirror of the reference's acceptance checks (the gate the host keeps).

  eval_residual_norm   verifier.f90:75-204   ||A v_j - l_j B v_j||_2 / ||A||_F, (avg, max)
  eval_orthogonality   verifier.f90:233-330  G = V^T B V, rows/cols scaled by 1/sqrt(G_jj),
                                             diagonal zeroed, Frobenius norm
  get_ipratios         distribute_matrix.f90:18-78  sum v^4 / (sum v (S v))^2

numpy on gathered arrays (1x1 grid); the normalisations are the reference's, including the
quirk that the orthogonality check normalises by the computed G_jj instead of comparing
with 1.
"""
import numpy as np


def eval_residual_norm(A, values, V, B=None):
    A = np.asarray(A)
    V = np.asarray(V)
    n_check = V.shape[1]
    a_norm = np.linalg.norm(A, "fro")
    R = (B @ V if B is not None else V.copy()) * (-np.asarray(values)[:n_check])
    R += A @ V
    norms = np.linalg.norm(R, axis=0)
    return a_norm, norms.sum() / a_norm / n_check, norms.max() / a_norm


def eval_orthogonality(V, B=None, index1=1, index2=None):
    V = np.asarray(V)
    index2 = V.shape[1] if index2 is None else index2
    Vs = V[:, index1 - 1:index2]
    G = Vs.T @ (B @ Vs if B is not None else Vs)
    s = 1.0 / np.sqrt(np.diag(G))
    G = G * s[:, None] * s[None, :]
    np.fill_diagonal(G, 0.0)
    return np.linalg.norm(G, "fro")


def get_ipratios(V, S=None):
    V = np.asarray(V)
    SV = S @ V if S is not None else V
    p4 = (V ** 4).sum(axis=0)
    p2 = (V * SV).sum(axis=0)
    return p4 / p2 ** 2
