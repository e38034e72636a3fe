"""Oracle for SURVEY.md 8 row f3: PartialCCA and GRCCA.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Reference-structured restatements
(cca_zoo/linear/_partialcca.py:67-103, 105-137 and cca_zoo/linear/_grcca.py:77-178: explicit n x d residuals /
augmented views fed to the MCCA hooks) and the second-moment forms the product implements.  Both are pinned by
``tests/golden/partial_group.npz`` (captured from the real reference).
"""

from __future__ import annotations

import numpy as np
import scipy.linalg

from oracle import gram_form as gf
from oracle import reference_form as rf


# ---------------------------------------------------------------- reference-structured
def _mcca_hooks(views, k, c, eps):
    """_partialcca.py:97-102 / _grcca.py:111-115: gevp(_build_A(views), _build_B(views, c), k), pca=False."""
    m = len(views)
    A = rf._between_view_cov(views)
    B = scipy.linalg.block_diag(*[(1.0 - c[i]) * np.atleast_2d(np.cov(v, rowvar=False)) + c[i] * np.eye(v.shape[1])
                                  for i, v in enumerate(views)])
    B = rf._eps_floor(B, eps) / m
    _, V = rf.top_eigenpairs(A, B, k)
    return np.split(V, np.cumsum([v.shape[1] for v in views])[:-1], axis=0)


def partialcca_reference_form(views, partials, k, c=0.0, eps=1e-6, center=True):
    vs, means = rf.center_views(views, center)
    Z = np.asarray(partials, dtype=float)
    betas = [np.linalg.pinv(Z) @ v for v in vs]                      # _partialcca.py:90-92
    resid = [v - Z @ b for v, b in zip(vs, betas)]                   # :93-95
    W = _mcca_hooks(resid, k, rf._per_view(c, 0.0, len(vs)), eps)
    return W, means, betas


def partialcca_transform(views, partials, W, means, betas):
    """_partialcca.py:121-137 (with partials) -- without partials it is rf.project."""
    Z = np.asarray(partials, dtype=float)
    return [((np.asarray(v) - mu) - Z @ b) @ w for v, mu, b, w in zip(views, means, betas, W)]


def _group_mean(arr, group):
    ids, inverse, counts = np.unique(group, return_inverse=True, return_counts=True)
    gm = np.array([arr[:, group == g].mean(axis=1) for g in ids]).T
    return inverse, counts, gm


def grcca_reference_form(views, groups, k, c=0.0, mu=0.0, eps=1e-6, center=True):
    vs, means = rf.center_views(views, center)
    m = len(vs)
    cs, mus = rf._per_view(c, 0.0, m), rf._per_view(mu, 0.0, m)
    aug = []
    for v, g, ci, mi in zip(vs, groups, cs, mus):                    # _grcca.py:127-141
        if ci <= 0:
            aug.append(v)
            continue
        inverse, counts, gm = _group_mean(v, g)
        mu_eff = 1.0 if mi == 0 else mi
        aug.append(np.hstack(((v - gm[:, inverse]) / ci, gm / np.sqrt(mu_eff / counts))))
    blocks = _mcca_hooks(aug, k, cs, eps)
    W = []
    for blk, g, ci, mi in zip(blocks, groups, cs, mus):              # _grcca.py:143-162
        if ci <= 0:
            W.append(blk)
            continue
        ng = np.unique(g).shape[0]
        w1, w2 = blk[:-ng], blk[-ng:]
        inverse, counts, gm = _group_mean(w1.T, g)
        mu_eff = 1.0 if mi == 0 else mi
        w1 = (w1 - gm[:, inverse].T) / ci
        w2 = w2 / np.sqrt(mu_eff * counts[:, None])
        W.append(w1 + w2[inverse])
    return W, means


# ---------------------------------------------------------------- from second moments
def partialcca_from_moments(G, s, n, dz, dims, k, c=None, eps=1e-6, center=True):
    """(G, s) are the moments of [Z | X_1 .. X_m]; returns weights, means, betas."""
    D = int(sum(dims))
    Gzz, Gzx, Gxx = G[:dz, :dz], G[:dz, dz:], G[dz:, dz:]
    s_z, s_x = s[:dz], s[dz:]
    mu = s_x / n if center else np.zeros(D)
    ZtXc = Gzx - np.outer(s_z, mu)
    beta = np.linalg.pinv(Gzz, hermitian=True) @ ZtXc
    XcXc = Gxx - np.outer(s_x, mu) - np.outer(mu, s_x) + n * np.outer(mu, mu)
    G_eff = XcXc - ZtXc.T @ beta
    s_eff = (s_x - n * mu) - s_z @ beta
    W, _, _ = gf.mcca_from_moments(G_eff, s_eff, n, dims, k, c=c, eps=eps, center=True)
    cuts = np.cumsum(dims)[:-1]
    return W, np.split(mu, cuts), np.split(beta, cuts, axis=1)


def grcca_from_moments(G, s, n, dims, groups, k, c=None, mu=None, eps=1e-6, center=True):
    m = len(dims)
    cs = [0.0] * m if c is None else list(c)
    mus = [0.0] * m if mu is None else list(mu)
    maps = []
    for d, g, ci, mi in zip(dims, groups, cs, mus):
        if ci <= 0:
            maps.append(np.eye(d))
            continue
        ids, inverse, counts = np.unique(g, return_inverse=True, return_counts=True)
        E = np.zeros((d, len(ids)))
        E[np.arange(d), inverse] = 1.0
        M = E / counts[None, :]
        mu_eff = 1.0 if mi == 0 else mi
        maps.append(np.hstack([(np.eye(d) - M @ E.T) / ci, M / np.sqrt(mu_eff / counts)[None, :]]))
    T = scipy.linalg.block_diag(*maps)
    dims_aug = [t.shape[1] for t in maps]
    blocks, _, _ = gf.mcca_from_moments(T.T @ G @ T, s @ T, n, dims_aug, k, c=cs, eps=eps, center=True)
    W = []
    for blk, g, ci, mi in zip(blocks, groups, cs, mus):
        if ci <= 0:
            W.append(blk)
            continue
        ids, inverse, counts = np.unique(g, return_inverse=True, return_counts=True)
        ng = len(ids)
        w1, w2 = blk[:-ng], blk[-ng:]
        gm = np.array([w1[g == gid].mean(axis=0) for gid in ids])
        mu_eff = 1.0 if mi == 0 else mi
        W.append((w1 - gm[inverse]) / ci + (w2 / np.sqrt(mu_eff * counts[:, None]))[inverse])
    means = np.split(s / n if center else np.zeros(int(sum(dims))), np.cumsum(dims)[:-1])
    return W, means
