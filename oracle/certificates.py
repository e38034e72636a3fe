"""Size-independent certificates: is a fitted solution THE solution the reference defines?

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  At BASELINE.json's full shapes (MCCA D = 8192, GCCA
D = 16384) a dense LAPACK eigen-solve of the oracle takes minutes, but every estimator on the hot path is the top-k
part of a symmetric-definite pencil ``A v = lam B v`` (``gevp``, cca_zoo/_utils/_linalg.py:44-73):

    rCCA / CCA / PLS   A = [[0, C12], [C21, 0]],  B = blockdiag(R_1, R_2),  v = [w_1; w_2] / sqrt(2),  lam = sigma
                       (cca_zoo/linear/_rcca.py:92-100 in whitened coordinates)
    MCCA               A = C - blockdiag(C_ii),    B = blockdiag(R_i + shift),  v = w / sqrt(m)
                       (cca_zoo/linear/_mcca.py:141-197, pencil (A/m, B/m))
    GCCA               A = Gx,                     B = blockdiag(R_i / mu_i),   v_i = mu_i R_i^-1 Gx_ii W_i / sqrt(lam)
                       (cca_zoo/linear/_gcca.py:95-109 in D x D Gram form, oracle.gram_form.gcca_from_moments)

and a candidate ``(lam, V)`` is the wanted eigen-system iff

    (1) the residual  A V - B V diag(lam)  vanishes,
    (2) V' B V = I,
    (3) exactly k eigenvalues of the pencil exceed ``lam_k (1 - delta)``   -- Sylvester's law of inertia on
        ``A - sigma B`` through one symmetric-indefinite factorization (LAPACK ``dsytrf``, ``_count_positive``).

(1)-(2) cost D^2 k flops, (3) one D^3/3 factorization: seconds where the full eigen-solve takes minutes.
``tests/test_round2_host.py`` checks on small problems that the oracle's own solutions pass and that perturbed ones
fail.
"""

from __future__ import annotations

import numpy as np
import scipy.linalg

from oracle.gram_form import _blocks, _eps_shift, covariance_from_moments

__all__ = ["pencil_certificate", "rcca_pencil", "mcca_pencil", "gcca_pencil"]


def _count_positive(M):
    """Number of positive eigenvalues of the symmetric matrix ``M``: inertia of its Bunch-Kaufman factor
    ``P L D L' P'`` (LAPACK ``dsytrf``; D has 1 x 1 and 2 x 2 diagonal blocks, a 2 x 2 pivot being indefinite by
    construction)."""
    from scipy.linalg import lapack

    M = np.asfortranarray(M, dtype=np.float64)
    lwork = int(lapack.dsytrf_lwork(M.shape[0], lower=1)[0])
    ldu, piv, info = lapack.dsytrf(M, lwork=lwork, lower=1, overwrite_a=1)
    if info < 0:
        raise ValueError(f"dsytrf: illegal argument {-info}")
    n = ldu.shape[0]
    dg = np.diagonal(ldu)
    sub = np.diagonal(ldu, -1)
    pos, i = 0, 0
    while i < n:
        if piv[i] < 0:                       # 2 x 2 block (i, i + 1)
            a, b, c = dg[i], sub[i], dg[i + 1]
            det, tr = a * c - b * b, a + c
            pos += 1 if det < 0 else (2 if tr > 0 else 0)
            i += 2
        else:
            pos += 1 if dg[i] > 0 else 0
            i += 1
    return pos


def pencil_certificate(A, B, V, lam, delta=1e-6, inertia=True):
    """Residuals of the three conditions for ``(lam (k,), V (D x k))`` on the dense pencil ``(A, B)``.

    Returns ``{"residual", "orthonormality", "n_above", "k"}``: relative Frobenius residual of (1), max-abs deviation
    of (2), and (if ``inertia``) the number of pencil eigenvalues above ``lam_k - delta * |lam_1|``."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    V = np.asarray(V, dtype=np.float64)
    lam = np.asarray(lam, dtype=np.float64)
    AV, BV = A @ V, B @ V
    out = {
        "residual": float(np.linalg.norm(AV - BV * lam[None, :]) / max(np.linalg.norm(BV * lam[None, :]), 1e-300)),
        "orthonormality": float(np.abs(V.T @ BV - np.eye(V.shape[1])).max()),
        "k": int(V.shape[1]),
    }
    if inertia:
        sigma = lam[-1] - delta * abs(lam[0])
        out["n_above"] = _count_positive(A - sigma * B)
    return out


def rcca_pencil(G, s, n, dims, c, center=True):
    """(A, B) of the two-view canonical ridge from the moments; ``V = [W1; W2] / sqrt(2)``, ``lam = sigma``."""
    b1, b2 = _blocks(dims)
    M = covariance_from_moments(G, s, n, center)
    D = int(sum(dims))
    A = np.zeros((D, D))
    A[b1, b2] = M[b1, b2]
    A[b2, b1] = M[b1, b2].T
    B = np.zeros((D, D))
    B[b1, b1] = (1.0 - c[0]) * M[b1, b1] + c[0] * np.eye(dims[0])
    B[b2, b2] = (1.0 - c[1]) * M[b2, b2] + c[1] * np.eye(dims[1])
    return A, B


def mcca_pencil(G, s, n, dims, c, eps=1e-6, shift=None):
    """(A, B) of MCCA from the moments (covariances always centred); ``V = W / sqrt(m)``, ``lam`` = eigenvalues.
    ``shift``: the eps-shift of the reference if already known (``None`` computes it from the blocks' spectra)."""
    bl = _blocks(dims)
    C = covariance_from_moments(G, s, n, True)
    D = int(sum(dims))
    A = C.copy()
    B = np.zeros((D, D))
    R = []
    for i, b in enumerate(bl):
        A[b, b] = 0.0
        R.append((1.0 - c[i]) * C[b, b] + c[i] * np.eye(dims[i]))
    if shift is None:
        shift = 0.0 if all(ci >= eps for ci in c) else _eps_shift(R, eps)
    for i, b in enumerate(bl):
        B[b, b] = R[i] + shift * np.eye(dims[i])
    return A, B


def gcca_pencil(G, s, n, dims, c, W, lam, view_weights=None, eps=1e-6, center=True, shifts=None):
    """(A, B, V) of GCCA in Gram form from the moments and a candidate ``(W, lam)``:
    ``A = Gx`` (second moments as fitted), ``B = blockdiag(R_i / mu_i)``, ``V_i = mu_i R_i^-1 Gx_ii W_i / sqrt(lam)``.
    ``shifts``: per-view eps-floors if already known (``None`` computes them; 0 when ``c_i >= eps``)."""
    m = len(dims)
    mu = [1.0] * m if view_weights is None else list(view_weights)
    bl = _blocks(dims)
    C = covariance_from_moments(G, s, n, True)
    Gx = (G - np.outer(s, s) / n) if center else np.asarray(G, dtype=np.float64)
    D = int(sum(dims))
    B = np.zeros((D, D))
    V = np.zeros((D, len(lam)))
    for i, b in enumerate(bl):
        R = (1.0 - c[i]) * C[b, b] + c[i] * np.eye(dims[i])
        sh = (0.0 if c[i] >= eps else _eps_shift([R], eps)) if shifts is None else shifts[i]
        R = R + sh * np.eye(dims[i])
        B[b, b] = R / mu[i]
        V[b] = mu[i] * scipy.linalg.cho_solve(scipy.linalg.cho_factor(R, lower=True), Gx[b, b] @ np.asarray(W[i], dtype=np.float64))
    V /= np.sqrt(np.asarray(lam, dtype=np.float64))[None, :]
    return Gx, B, V
