"""Oracle layer 1: the reference algorithms, structured the way the reference is.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Plain functions over
NumPy arrays; no estimator classes.  File:line citations are relative to
``/root/reference``.  Pinned by ``tests/golden/*.npz`` (outputs of the real
reference captured by ``tools/gen_golden.py``).
"""

from __future__ import annotations

import numpy as np
import scipy.linalg

__all__ = [
    "center_views",
    "thin_svd_whitener",
    "top_eigenpairs",
    "rcca_weights",
    "mcca_weights",
    "gcca_weights",
    "project",
    "pairwise_corr",
    "mean_offdiag_corr",
    "factor_loadings",
    "joint_data",
]


# --------------------------------------------------------------------------
# cca_zoo/_base.py:78-102  (BaseModel._setup_fit)
# --------------------------------------------------------------------------
def center_views(views, center=True):
    """Column means and (optionally) centred copies.

    Mirrors ``_setup_fit``: with ``center=False`` the stored means are zeros of
    dtype float64 (``np.zeros(p)``, _base.py:101) and data are left untouched.
    """
    views = [np.asarray(v) for v in views]
    if center:
        means = [v.mean(axis=0) for v in views]
        return [v - m for v, m in zip(views, means)], means
    return views, [np.zeros(v.shape[1]) for v in views]


# --------------------------------------------------------------------------
# cca_zoo/_utils/_linalg.py:9-41  (svd_whiten)
# --------------------------------------------------------------------------
def thin_svd_whitener(X, ridge=0.0):
    """Return ``(X_white, W)`` from a thin SVD of the n x d data matrix.

    ``lam = s**2/(n-1)``; ``W = V * ((1-ridge)*lam + ridge)**-0.5``; directions
    with ``s <= 0`` *exactly* are dropped (_linalg.py:30-33).
    """
    n = X.shape[0]
    U, s, Vt = np.linalg.svd(X, full_matrices=False)
    keep = s > 0
    U, s, Vt = U[:, keep], s[keep], Vt[keep]
    lam = s * s / (n - 1)
    scale = 1.0 / np.sqrt((1.0 - ridge) * lam + ridge)
    return U * (s * scale), Vt.T * scale


# --------------------------------------------------------------------------
# cca_zoo/_utils/_linalg.py:44-73  (gevp)
# --------------------------------------------------------------------------
def top_eigenpairs(A, B, k):
    """k largest eigenpairs of ``A v = lam B v`` (B may be None), descending.

    scipy picks LAPACK syevr (B None) / sygvx (B given) because a subset is
    requested; sygvx normalises ``v' B v = 1``.
    """
    p = A.shape[0]
    k = min(k, p)
    if B is None:
        w, V = scipy.linalg.eigh(A, subset_by_index=[p - k, p - 1])
    else:
        w, V = scipy.linalg.eigh(A, B, subset_by_index=[p - k, p - 1])
    order = np.argsort(w)[::-1]
    return w[order].real, V[:, order].real


def _per_view(value, default, m):
    # cca_zoo/_utils/_validation.py:45-75 (perview_parameter)
    if value is None:
        return [default] * m
    if isinstance(value, (list, tuple)):
        if len(value) != m:
            raise ValueError("per-view parameter has wrong length")
        return list(value)
    return [value] * m


# --------------------------------------------------------------------------
# cca_zoo/linear/_rcca.py:69-101  (rCCA.fit; CCA = c 0, PLS = c 1)
# --------------------------------------------------------------------------
def rcca_weights(views, k, c=0.0, center=True):
    """Weights and means of the two-view canonical ridge, reference-structured."""
    if len(views) != 2:
        raise ValueError("rCCA requires exactly 2 views")
    (X1, X2), means = center_views(views, center)
    c1, c2 = _per_view(c, 0.0, 2)
    X1w, W1 = thin_svd_whitener(X1, c1)
    X2w, W2 = thin_svd_whitener(X2, c2)
    k = min(k, X1w.shape[1], X2w.shape[1])
    T = X1w.T @ X2w / (X1.shape[0] - 1)
    U, _, Vt = np.linalg.svd(T, full_matrices=False)
    return [W1 @ U[:, :k], W2 @ Vt[:k].T], means


# --------------------------------------------------------------------------
# cca_zoo/linear/_mcca.py:99-197  (MCCA.fit, _build_A, _build_B, _build_B_pca)
# --------------------------------------------------------------------------
def _pca_full(v):
    """What ``sklearn.decomposition.PCA().fit(v)`` provides to MCCA.

    components (rows), explained variances (ddof=1) and the projection of the
    *re-centred* data.  Component signs are arbitrary and cancel in the final
    weights (``components.T @ w``).
    """
    mu = v.mean(axis=0)
    vc = v - mu
    _, s, Vt = np.linalg.svd(vc, full_matrices=False)
    return Vt, s * s / (v.shape[0] - 1), vc @ Vt.T


def _between_view_cov(views):
    # _mcca.py:141-153 : np.cov of the stacked views minus its diagonal blocks
    A = np.cov(np.hstack(views), rowvar=False)
    A -= scipy.linalg.block_diag(*[np.atleast_2d(np.cov(v, rowvar=False)) for v in views])
    return A / len(views)


def _eps_floor(B, eps):
    # _mcca.py:170-172 / 194-196 and _gcca.py:102-104
    lo = np.linalg.eigvalsh(B).min()
    if lo < eps:
        B = B + (eps - lo) * np.eye(B.shape[0])
    return B


def mcca_weights(views, k, c=0.0, pca=True, eps=1e-6, center=True):
    """Weights and means of multiset CCA, reference-structured."""
    vs, means = center_views(views, center)
    m = len(vs)
    cs = _per_view(c, 0.0, m)
    if pca:
        fits = [_pca_full(v) for v in vs]
        proj = [f[2] for f in fits]
        A = _between_view_cov(proj)
        B = scipy.linalg.block_diag(
            *[np.diag((1.0 - cs[i]) * fits[i][1] + cs[i]) for i in range(m)]
        )
        B = _eps_floor(B, eps) / m
        widths = [p.shape[1] for p in proj]
    else:
        A = _between_view_cov(vs)
        B = scipy.linalg.block_diag(
            *[
                (1.0 - cs[i]) * np.atleast_2d(np.cov(v, rowvar=False)) + cs[i] * np.eye(v.shape[1])
                for i, v in enumerate(vs)
            ]
        )
        B = _eps_floor(B, eps) / m
        widths = [v.shape[1] for v in vs]
    _, V = top_eigenpairs(A, B, k)
    parts = np.split(V, np.cumsum(widths)[:-1], axis=0)
    if pca:
        parts = [fits[i][0].T @ parts[i] for i in range(m)]
    return parts, means


# --------------------------------------------------------------------------
# cca_zoo/linear/_gcca.py:80-110  (GCCA.fit) -- the n x n formulation
# --------------------------------------------------------------------------
def gcca_weights(views, k, c=0.0, view_weights=None, eps=1e-6, center=True):
    """Weights and means of generalised CCA via the n x n matrix Q."""
    vs, means = center_views(views, center)
    m = len(vs)
    cs = _per_view(c, 0.0, m)
    mu = _per_view(view_weights, 1.0, m)
    n = vs[0].shape[0]
    Q = np.zeros((n, n))
    for v, ci, mi in zip(vs, cs, mu):
        R = (1.0 - ci) * np.atleast_2d(np.cov(v, rowvar=False)) + ci * np.eye(v.shape[1])
        R = _eps_floor(R, eps)
        Q += mi * (v @ np.linalg.inv(R) @ v.T)
    _, T = top_eigenpairs(Q, None, k)
    T = T[:, :k]
    return [np.linalg.pinv(v) @ T for v in vs], means


# --------------------------------------------------------------------------
# cca_zoo/_base.py:108-234  (transform / pairwise_correlations / score / loadings)
# --------------------------------------------------------------------------
def project(views, weights, means):
    return [(np.asarray(v) - mu) @ w for v, mu, w in zip(views, means, weights)]


def pairwise_corr(views, weights, means):
    """(m, m, k) Pearson correlations between the canonical variates."""
    T = np.stack(project(views, weights, means), axis=0)
    T = T - T.mean(axis=1, keepdims=True)
    nrm = np.sqrt((T * T).sum(axis=1, keepdims=True))
    T = T / np.where(nrm > 1e-12, nrm, 1.0)
    return np.einsum("isd,jsd->ijd", T, T)


def mean_offdiag_corr(views, weights, means):
    """``score``: mean of the off-diagonal pairwise correlations, per dimension."""
    R = pairwise_corr(views, weights, means)
    m = R.shape[0]
    off = R.sum(axis=(0, 1)) - sum(R[i, i] for i in range(m))
    return off / (m * (m - 1))


def factor_loadings(views, weights, means):
    out = []
    for v, t in zip(views, project(views, weights, means)):
        v = np.asarray(v)
        vc = v - v.mean(axis=0)
        tc = t - t.mean(axis=0)
        cov = vc.T @ tc / (v.shape[0] - 1)
        sv = np.maximum(vc.std(axis=0, ddof=1), 1e-12)
        st = np.maximum(tc.std(axis=0, ddof=1), 1e-12)
        out.append(cov / np.outer(sv, st))
    return out


# --------------------------------------------------------------------------
# cca_zoo/datasets/_simulated.py:49-130  (JointData)
# --------------------------------------------------------------------------
def joint_data(n_views=2, n_samples=100, latent_dimensions=1, n_features=10,
               signal_to_noise=1.0, random_state=None, n_draws=1):
    """Reproduce ``JointData(...).sample()`` including the RNG draw order.

    One ``default_rng(random_state)``; loadings ``W_i`` (d_i x k) drawn at
    construction in view order; each ``sample()`` draws ``z`` then, per view,
    the noise.  Returns the list of draws (each a list of views) when
    ``n_draws > 1``.
    """
    rng = np.random.default_rng(random_state)
    feats = _per_view(n_features, 10, n_views)
    snrs = _per_view(signal_to_noise, 1.0, n_views)
    loadings = [rng.standard_normal((p, latent_dimensions)) for p in feats]
    draws = []
    for _ in range(n_draws):
        z = rng.standard_normal((n_samples, latent_dimensions))
        vs = []
        for w, snr in zip(loadings, snrs):
            sig = z @ w.T
            sd = 1.0 / np.sqrt(snr) if snr > 0 else 1.0
            vs.append(sig + rng.standard_normal(sig.shape) * sd)
        draws.append(vs)
    return draws[0] if n_draws == 1 else draws
