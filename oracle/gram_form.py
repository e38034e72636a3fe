"""Oracle layer 2: the hot path restated from second moments only.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  This is the dense
NumPy float64 *specification* of what ``libccz`` computes on the device:

    one pass over the rows  ->  G = [X_1..X_m]'[X_1..X_m]  (D x D),  s = 1'[X]  (D)
    C = (G - s s'/n)/(n-1)                     (centred covariance, never materialising X - mean)
    R_i = (1-c_i) C_ii + c_i I = L_i L_i'      (Cholesky whitening instead of an n x d SVD)
    small dense top-k eigen / singular problems in the whitened coordinates

and is checked against ``oracle.reference_form`` and the goldens captured from
the real reference (SURVEY.md section 8(a) "Verified semantics").  The device
path uses a block-Krylov top-k solver where this file calls ``eigh``/``svd``;
that is an implementation detail with the same mathematical result.
"""

from __future__ import annotations

import numpy as np
import scipy.linalg

__all__ = [
    "moments",
    "covariance_from_moments",
    "rcca_from_moments",
    "mcca_from_moments",
    "gcca_from_moments",
    "whitener_from_gram",
]

_RANK_TOL = 64.0 * np.finfo(np.float64).eps  # relative eigenvalue floor of the fallback


def moments(views):
    """Raw second moments and column sums of the column-stacked views (float64)."""
    X = np.hstack([np.asarray(v, dtype=np.float64) for v in views])
    return X.T @ X, X.sum(axis=0), X.shape[0]


def covariance_from_moments(G, s, n, center=True):
    """``(G - s s'/n)/(n-1)`` if ``center`` else ``G/(n-1)`` (reference: _base.py:97-101)."""
    if center:
        return (G - np.outer(s, s) / n) / (n - 1)
    return G / (n - 1)


def _blocks(dims):
    edges = np.concatenate([[0], np.cumsum(dims)])
    return [slice(int(edges[i]), int(edges[i + 1])) for i in range(len(dims))]


def _chol_or_none(R):
    try:
        return np.linalg.cholesky(R)
    except np.linalg.LinAlgError:
        return None


def _floored_whitener(R):
    """Fallback when R is not numerically SPD: drop eigen-directions below a
    relative floor (the reference keeps ~1e-15 junk directions there and its
    output is noise -- SURVEY.md 8(a) "d > n ... c = 0")."""
    lam, V = np.linalg.eigh(R)
    keep = lam > _RANK_TOL * R.shape[0] * max(lam.max(), 0.0)
    return V[:, keep] / np.sqrt(lam[keep])


def _svd_topk(T, k):
    """Leading k singular triplets of T through ``scipy.linalg.eigh(subset_by_index)`` of the smaller Gram side
    (minutes -> seconds at d = 4096, k = 64); same result as the dense SVD when sigma_k is well above round-off."""
    p, q = T.shape
    if p <= q:
        lam, U = scipy.linalg.eigh(T @ T.T, subset_by_index=[p - k, p - 1])
        U = U[:, ::-1]
        sv = np.sqrt(np.maximum(lam[::-1], 0.0))
        return U, sv, ((T.T @ U) / sv).T
    lam, V = scipy.linalg.eigh(T.T @ T, subset_by_index=[q - k, q - 1])
    V = V[:, ::-1]
    sv = np.sqrt(np.maximum(lam[::-1], 0.0))
    return (T @ V) / sv, sv, V.T


def rcca_from_moments(G, s, n, dims, k, c=(0.0, 0.0), center=True, fast=False):
    """rCCA / CCA / PLS weights (cca_zoo/linear/_rcca.py:69-101) from (G, s, n).

    ``center=False`` uses the *uncentred* second moments (the reference SVDs
    the raw data then).  Each weight column satisfies ``w' R_i w = 1``.
    ``fast=True`` takes only the leading k singular triplets (LAPACK syevr on the Gram side) instead of the full
    SVD -- for the BASELINE shapes (d = 4096) where the dense SVD takes a minute.
    """
    b1, b2 = _blocks(dims)
    M = covariance_from_moments(G, s, n, center)
    R1 = (1.0 - c[0]) * M[b1, b1] + c[0] * np.eye(dims[0])
    R2 = (1.0 - c[1]) * M[b2, b2] + c[1] * np.eye(dims[1])
    k = min(k, dims[0], dims[1], n)
    L1, L2 = _chol_or_none(R1), _chol_or_none(R2)
    if L1 is not None and L2 is not None:
        T = scipy.linalg.solve_triangular(L1, M[b1, b2], lower=True)
        T = scipy.linalg.solve_triangular(L2, T.T, lower=True).T        # L1^-1 M12 L2^-T
        U, sv, Vt = _svd_topk(T, k) if fast else np.linalg.svd(T, full_matrices=False)
        W1 = scipy.linalg.solve_triangular(L1.T, U[:, :k], lower=False)
        W2 = scipy.linalg.solve_triangular(L2.T, Vt[:k].T, lower=False)
    else:
        P1, P2 = _floored_whitener(R1), _floored_whitener(R2)
        k = min(k, P1.shape[1], P2.shape[1])
        U, sv, Vt = np.linalg.svd(P1.T @ M[b1, b2] @ P2, full_matrices=False)
        W1, W2 = P1 @ U[:, :k], P2 @ Vt[:k].T
    means = [s[b] / n if center else np.zeros(d) for b, d in zip((b1, b2), dims)]
    return [W1, W2], means, sv[:k]


def _eps_shift(R_blocks, eps):
    """Reference rule (_mcca.py:170-172): shift = eps - min eig if min eig < eps."""
    lo = min(np.linalg.eigvalsh(R).min() for R in R_blocks)
    return (eps - lo) if lo < eps else 0.0


def mcca_from_moments(G, s, n, dims, k, c=None, eps=1e-6, center=True, fast=False):
    """MCCA weights (cca_zoo/linear/_mcca.py:99-197) from (G, s, n).

    Covariances are always centred (``np.cov`` / ``PCA`` re-centre even when
    ``center=False``); ``pca`` does not change the result (orthogonal change of
    basis), so it is not a parameter here.  Normalisation ``v' (B/m) v = 1``.
    """
    m = len(dims)
    c = [0.0] * m if c is None else list(c)
    bl = _blocks(dims)
    C = covariance_from_moments(G, s, n, True)
    R = [(1.0 - c[i]) * C[bl[i], bl[i]] + c[i] * np.eye(dims[i]) for i in range(m)]
    # (1-c) C + c I with C >= 0 has min eig >= c: no shift possible when every c_i >= eps (saves m dense eigvalsh)
    shift = 0.0 if all(ci >= eps for ci in c) else _eps_shift(R, eps)
    L = [np.linalg.cholesky(R[i] + shift * np.eye(dims[i])) for i in range(m)]
    D = int(sum(dims))
    S = np.zeros((D, D))
    for i in range(m):
        for j in range(m):
            if i == j:
                continue
            t = scipy.linalg.solve_triangular(L[i], C[bl[i], bl[j]], lower=True)
            S[bl[i], bl[j]] = scipy.linalg.solve_triangular(L[j], t.T, lower=True).T
    k = min(k, D)
    if fast:      # leading k eigenpairs only (LAPACK syevr): the dense solve takes minutes at D = 8192
        lam, Y = scipy.linalg.eigh(S, subset_by_index=[D - k, D - 1])
        lam, Y = lam[::-1], Y[:, ::-1]
    else:
        lam, Y = np.linalg.eigh(S)
        lam, Y = lam[::-1][:k], Y[:, ::-1][:, :k]
    W = [np.sqrt(m) * scipy.linalg.solve_triangular(L[i].T, Y[bl[i]], lower=False) for i in range(m)]
    means = [s[b] / n if center else np.zeros(d) for b, d in zip(bl, dims)]
    return W, means, lam  # eigenvalues of (A/m, B/m) equal those of (A, B)


def _pinv_sym(Gii, rhs):
    L = _chol_or_none(Gii)
    if L is not None:
        return scipy.linalg.cho_solve((L, True), rhs)
    lam, V = np.linalg.eigh(Gii)
    keep = lam > _RANK_TOL * Gii.shape[0] * max(lam.max(), 0.0)
    return (V[:, keep] / lam[keep]) @ (V[:, keep].T @ rhs)


def gcca_from_moments(G, s, n, dims, k, c=None, view_weights=None, eps=1e-6, center=True, topk=None):
    """GCCA weights (cca_zoo/linear/_gcca.py:80-110) in D x D Gram form.

    ``R_i`` always from the centred covariance (+ per-view eps floor);
    ``K_ij = sqrt(mu_i mu_j) L_i^-1 Gx_ij L_j^-T`` with ``Gx`` the second
    moments of the data *as fitted* (centred iff ``center``); top-k
    ``K u = lam u``; ``W_i = Gx_ii^+ sum_j sqrt(mu_j) Gx_ij L_j^-T u_j / sqrt(lam)``.
    ``topk(K, k)``: optional replacement of the dense ``eigh`` for the top-k pairs (Lanczos at D = 16384).
    """
    m = len(dims)
    c = [0.0] * m if c is None else list(c)
    mu = [1.0] * m if view_weights is None else list(view_weights)
    bl = _blocks(dims)
    C = covariance_from_moments(G, s, n, True)
    Gx = (G - np.outer(s, s) / n) if center else G
    L = []
    for i in range(m):
        R = (1.0 - c[i]) * C[bl[i], bl[i]] + c[i] * np.eye(dims[i])
        R = R + _eps_shift([R], eps) * np.eye(dims[i])
        L.append(np.linalg.cholesky(R))
    D = int(sum(dims))
    # Z_j = sqrt(mu_j) * Gx[:, j] L_j^-T   (D x d_j): right half of K and of the back-projection
    Z = np.zeros((D, D))
    for j in range(m):
        Z[:, bl[j]] = np.sqrt(mu[j]) * scipy.linalg.solve_triangular(L[j], Gx[:, bl[j]].T, lower=True).T
    K = np.zeros((D, D))
    for i in range(m):
        K[bl[i], :] = np.sqrt(mu[i]) * scipy.linalg.solve_triangular(L[i], Z[bl[i], :], lower=True)
    K = 0.5 * (K + K.T)
    k = min(k, D, n)
    if topk is None:
        lam, Uv = np.linalg.eigh(K)
        lam, Uv = lam[::-1][:k], Uv[:, ::-1][:, :k]
    else:                                            # a sparse top-k solver for sizes where the dense eigh takes minutes
        lam, Uv = topk(K, k)                         # (descending eigenvalues (k,), eigenvectors (D, k))
    rhs = Z @ Uv / np.sqrt(lam)                      # X' T  stacked by view   (D x k)
    W = [_pinv_sym(Gx[bl[i], bl[i]], rhs[bl[i]]) for i in range(m)]
    means = [s[b] / n if center else np.zeros(d) for b, d in zip(bl, dims)]
    return W, means, lam


def whitener_from_gram(Gxx, n, ridge=0.0):
    """``svd_whiten`` (cca_zoo/_utils/_linalg.py:9-41) from the Gram X'X of a
    centred view: eigenvalues ``lam`` of ``X'X/(n-1)`` descending (the top
    ``min(n, d)`` kept) and ``W = V ((1-ridge) lam + ridge)^-1/2``."""
    d = Gxx.shape[0]
    lam, V = np.linalg.eigh(Gxx / (n - 1))
    lam, V = lam[::-1], V[:, ::-1]
    r = min(n, d)
    lam = np.maximum(lam[:r], 0.0)
    return V[:, :r] / np.sqrt((1.0 - ridge) * lam + ridge), lam
