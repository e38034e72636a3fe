"""Group-regularised CCA from second moments (reference: cca_zoo/linear/_grcca.py:18-200).

The reference augments every (centred) view with per-group mean features,

    X_aug = [ (X - X M E') / c  |  X M diag(1 / sqrt(mu / counts)) ]  =  X T,

(``E`` the d x g membership indicator, ``M = E diag(1/counts)``), builds the MCCA matrices from the n x (d + g)
augmented views and collapses the eigenvectors back (``_collapse_weights``).  ``X_aug = X T`` is linear in X, so
the augmented views' moments are ``T' G T`` and ``T' s``: K1 runs on the ORIGINAL views, the augmentation is two
device GEMMs on the D x D moments, and the eigenproblem is the ordinary ``ccz_mcca_solve``.
"""

from __future__ import annotations

import warnings

import numpy as np

from cca_zoo_amd import _backend
from cca_zoo_amd._moments import compute_moments
from cca_zoo_amd._utils._validation import perview_parameter
from cca_zoo_amd.linear._mcca import MCCA


def _group_maps(group):
    ids, inverse, counts = np.unique(np.asarray(group), return_inverse=True, return_counts=True)
    d, g = len(inverse), len(ids)
    E = np.zeros((d, g))
    E[np.arange(d), inverse] = 1.0
    return E, inverse, counts


def _augment_map(group, c, mu):
    """``T`` with ``X_aug = X T`` (d x (d + g)); the identity when ``c <= 0`` (reference: ``_augment_view``)."""
    d = len(group)
    if c <= 0:
        return np.eye(d)
    E, _, counts = _group_maps(group)
    M = E / counts[None, :]
    mu_eff = 1.0 if mu == 0 else mu
    return np.hstack([(np.eye(d) - M @ E.T) / c, M / np.sqrt(mu_eff / counts)[None, :]])


def _collapse(block, group, c, mu):
    """Augmented-space eigenvector block -> original feature space (reference: ``_collapse_weights``)."""
    if c <= 0:
        return block
    E, inverse, counts = _group_maps(group)
    g = E.shape[1]
    w1, w2 = block[:-g], block[-g:]
    mu_eff = 1.0 if mu == 0 else mu
    w1 = (w1 - (E @ ((E.T @ w1) / counts[:, None]))) / c
    w2 = w2 / np.sqrt(mu_eff * counts[:, None])
    return w1 + w2[inverse]


class GRCCA(MCCA):
    """Group-regularised CCA (Tuzhilina et al.): within-group shrinkage ``c`` and group-mean penalty ``mu`` per view;
    ``fit(views, feature_groups=[labels_1, ..])`` with integer group labels per feature."""

    _parameter_constraints = {**MCCA._parameter_constraints}

    def __init__(self, latent_dimensions: int = 1, center: bool = True, c: float | list[float] = 0.0,
                 mu: float | list[float] = 0.0, eps: float = 1e-6) -> None:
        super().__init__(latent_dimensions=latent_dimensions, center=center, c=c, pca=False, eps=eps)
        self.mu = mu

    def fit(self, views, y=None, feature_groups=None):
        views_ = self._setup_fit(views)
        c_ = perview_parameter("c", self.c, 0.0, self.n_views_)
        mu_ = perview_parameter("mu", self.mu, 0.0, self.n_views_)
        if feature_groups is None:
            if any(ci > 0 for ci in c_):
                warnings.warn("No feature_groups provided; using a single group per view, which makes the group "
                              "regularisation a no-op.")
            feature_groups = [np.ones(int(v.shape[1]), dtype=int) for v in views_]
        self.feature_groups_ = feature_groups
        h = _backend.handle_for(views_)
        mom, keep, n, dims, kind = compute_moments(views_, h)
        self.n_samples_ = int(n)
        D = int(sum(dims))
        maps = [_augment_map(np.asarray(g), ci, mi) for g, ci, mi in zip(feature_groups, c_, mu_)]
        dims_aug = [int(T.shape[1]) for T in maps]
        Da = int(sum(dims_aug))
        T = np.zeros((D, Da))
        o = oa = 0
        for d, da, Ti in zip(dims, dims_aug, maps):
            T[o:o + d, oa:oa + da] = Ti
            o, oa = o + d, oa + da
        s = h.to_host(mom, (D,), offset_bytes=D * D * 8)
        mu_x = s / n if self.center else np.zeros(D)
        # moments of the augmented, centred views: T' (G - n mu mu') T and (s - n mu)' T  (np.cov re-centres anyway)
        h.moments_symmetrize(mom, D)
        Td = h.to_device(T)
        GT = h.alloc(D * Da * 8)
        aug = h.alloc((Da * Da + Da) * 8)
        h.gemm(0, 0, D, Da, D, 1.0, mom, D, Td.ptr, Da, 0.0, GT.ptr, Da)
        h.gemm(1, 0, Da, Da, D, 1.0, Td.ptr, Da, GT.ptr, Da, 0.0, aug.ptr, Da)
        h.h2d(aug.ptr + Da * Da * 8, (s[:, None] * T).sum(axis=0))
        W, _, vals = h.mcca_solve(aug.ptr, n, dims_aug, c_, self.eps, True, self.latent_dimensions)
        W = [_collapse(w, np.asarray(g), ci, mi) for w, g, ci, mi in zip(W, feature_groups, c_, mu_)]
        self._store(W, np.split(mu_x, np.cumsum(dims)[:-1]), kind, weights_like_input=False)
        self.eigenvalues_ = vals
        del keep
        return self
