"""GCCA -- generalised CCA in D x D Gram form on the device.

Reference: cca_zoo/linear/_gcca.py:80-110 builds the n x n matrix
``Q = sum_i mu_i X_i R_i^-1 X_i'`` (O(n^2 d); infeasible beyond n ~ 5e4), takes its
top-k eigenvectors T and sets ``W_i = pinv(X_i) T``.  The same weights follow from
second moments only (verified <= 7e-14): with
``M = [sqrt(mu_i) X_i L_i^-T]`` (``R_i = L_i L_i'``), ``Q = M M'`` shares its non-zero
spectrum with the D x D matrix ``K = M'M``; ``T = M u / sqrt(lam)`` and
``W_i = (X_i'X_i)^+ sum_j sqrt(mu_j) X_i'X_j L_j^-T u_j / sqrt(lam)``
(``ccz_gcca_solve``).  ``R_i`` always uses the centred covariance (``np.cov``) with the
per-view ``eps`` floor; the cross moments follow ``center``.
"""

from __future__ import annotations

import time
from typing import Any, ClassVar

import numpy as np

from cca_zoo_amd import _backend, _moments
from cca_zoo_amd._base import BaseModel
from cca_zoo_amd._moments import compute_moments
from cca_zoo_amd._utils._param_constraints import POSITIVE_EPS, RIDGE_PARAMETER
from cca_zoo_amd._utils._validation import perview_parameter


class GCCA(BaseModel):
    """Generalised CCA (MAX-VAR) for two or more views.

    Args:
        latent_dimensions: number of latent dimensions (default 1).
        center: subtract column means before fitting.
        c: ridge parameter(s) in ``[0, 1]``.
        view_weights: per-view weights ``mu_i`` (default all 1).
        eps: regularisation floor of the within-view matrices.
    """

    _parameter_constraints: ClassVar[dict[str, list[Any]]] = {
        **BaseModel._parameter_constraints,
        "c": RIDGE_PARAMETER,
        "eps": POSITIVE_EPS,
    }

    def __init__(self, latent_dimensions: int = 1, center: bool = True, c: float | list[float] = 0.0,
                 view_weights: list[float] | None = None, eps: float = 1e-6) -> None:
        super().__init__(latent_dimensions=latent_dimensions, center=center)
        self.c = c
        self.view_weights = view_weights
        self.eps = eps

    def _fit_moments(self, h, mom, n_total, dims, kind) -> None:
        c_ = perview_parameter("c", self.c, 0.0, self.n_views_)
        mu = perview_parameter("view_weights", self.view_weights, 1.0, self.n_views_)
        W, means, vals = h.gcca_solve(mom, n_total, dims, c_, mu, self.eps, self.center, self.latent_dimensions)
        # the reference takes the top latent_dimensions eigenvectors of an n x n matrix of rank <= D: beyond D they
        # span its null space and pinv(X_i) maps them to zero -- same shape here: zero columns up to min(k, n)
        k_ref = int(min(self.latent_dimensions, n_total))
        if W[0].shape[1] < k_ref:
            pad = k_ref - W[0].shape[1]
            W = [np.hstack([w, np.zeros((w.shape[0], pad))]) for w in W]
            vals = np.concatenate([vals, np.zeros(pad)])
        self._store(W, means, kind, weights_like_input=False)
        self.eigenvalues_ = vals

    def fit(self, views, y=None):
        views_ = self._setup_fit(views)
        h = _backend.handle_for(views_)
        t0 = time.perf_counter()
        mom, keep, n_total, dims, kind = compute_moments(views_, h, defer_offdiag=True)
        t1 = time.perf_counter()
        self.n_samples_ = int(n_total)            # inside row_sharded(): the global row count
        try:
            self._fit_moments(h, mom, n_total, dims, kind)
        except BaseException:
            _moments.settle_deferred(h)       # parameter / solver errors: the exchange's deferred half must not outlive `mom`
            raise
        # wall-clock split of this fit (K1 incl. the all-reduce | the d x d solve incl. the copy of the weights to the host)
        self.timings_ = {"moments_ms": (t1 - t0) * 1e3, "allreduce_ms": _moments.LAST["allreduce_ms"],
                         "solve_ms": (time.perf_counter() - t1) * 1e3,
                         "k1_route": h.moments_last_route()[0]}
        del keep
        return self
