"""MCCA -- multiset CCA as a generalised eigenproblem on the device.

Reference: cca_zoo/linear/_mcca.py:99-197.  The reference runs a full PCA (SVD) per
view, ``np.cov`` of the stacked views and LAPACK ``sygvx``.  Here: one MFMA pass for
all second moments; ``B = blockdiag((1-c_i) C_ii + c_i I)`` (+ the reference's
``eps - min_eig`` shift rule) is Cholesky-factored per block; the top-k eigenpairs of
``L^-1 (C - blockdiag C) L^-T`` are found by Chebyshev-filtered subspace iteration and
back-projected with the ``v'(B/m)v = 1`` normalisation (``ccz_mcca_solve``).

``pca`` is accepted and validated for API compatibility; PCA is an orthogonal change
of basis that leaves the weights unchanged (verified to 5e-14), so the
device path does not need it.  Covariances are always centred (``np.cov``/``PCA``
re-centre even when ``center=False``); ``means_`` still follow ``center``.

``_build_A`` / ``_build_B`` remain as overridable hooks with the reference's
semantics (used by subclasses that pre-transform views); they evaluate through the
same device moments.
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


class MCCA(BaseModel):
    """Multiset CCA for two or more views.

    Args:
        latent_dimensions: number of latent dimensions (default 1).
        center: subtract column means before fitting.
        c: ridge parameter(s) in ``[0, 1]`` (scalar or one per view).
        pca: accepted for compatibility (see module docstring).
        eps: positive-definiteness floor applied to ``B``.
    """

    _parameter_constraints: ClassVar[dict[str, list[Any]]] = {
        **BaseModel._parameter_constraints,
        "c": RIDGE_PARAMETER,
        "pca": ["boolean"],
        "eps": POSITIVE_EPS,
    }

    def __init__(self, latent_dimensions: int = 1, center: bool = True, c: float | list[float] = 0.0,
                 pca: bool = True, eps: float = 1e-6) -> None:
        super().__init__(latent_dimensions=latent_dimensions, center=center)
        self.c = c
        self.pca = pca
        self.eps = eps

    def _fit_moments(self, h, mom, n_total, dims, kind) -> None:
        c_ = perview_parameter("c", self.c, 0.0, self.n_views_)
        W, means, vals = h.mcca_solve(mom, n_total, dims, c_, self.eps, self.center, self.latent_dimensions)
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

    # -- hooks with the reference's semantics (cca_zoo/linear/_mcca.py:141-173) ----------
    def _covariance(self, views):
        h = _backend.default_handle()
        mom, keep, n, dims, _ = compute_moments([np.asarray(v) for v in views], h)
        D = sum(dims)
        h.moments_symmetrize(mom, D)          # K1 fills upper-triangular tiles only
        flat = h.to_host(mom, (D * D + D,))
        G, s = flat[: D * D].reshape(D, D), flat[D * D:]
        return (G - np.outer(s, s) / n) / (n - 1), dims

    def _build_A(self, views):
        C, dims = self._covariance(views)
        o = 0
        for d in dims:
            C[o:o + d, o:o + d] = 0.0
            o += d
        return C / len(views)

    def _build_B(self, views, c):
        C, dims = self._covariance(views)
        B = np.zeros_like(C)
        o = 0
        for i, d in enumerate(dims):
            B[o:o + d, o:o + d] = (1.0 - c[i]) * C[o:o + d, o:o + d] + c[i] * np.eye(d)
            o += d
        from cca_zoo_amd._utils._linalg import gevp

        lo = -gevp(-B, None, 1)[0][0]          # smallest eigenvalue via the device solver
        if lo < self.eps:
            B += (self.eps - lo) * np.eye(B.shape[0])
        return B / len(views)
