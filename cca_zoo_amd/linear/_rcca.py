"""rCCA / CCA / PLS -- two-view canonical ridge on the device.

Reference: cca_zoo/linear/_rcca.py:69-101 (``rCCA.fit``), _cca.py:42-76 (``c=0``),
_pls.py:42-77 (``c=1``).  The reference whitens each view with a thin SVD of the
n x d data and SVDs the whitened cross-covariance; here one MFMA pass builds the
second moments, ``R_i = (1-c_i) C_ii + c_i I`` is Cholesky-factored on the device and
the top-k singular triplets of ``L_1^-1 C_12 L_2^-T`` are back-projected
(``ccz_rcca_solve``).  Same weights up to column signs; each column satisfies
``w' R_i w = 1``.
"""

from __future__ import annotations

import time
from typing import Any, ClassVar

from cca_zoo_amd import _backend, _moments
from cca_zoo_amd._base import BaseModel
from cca_zoo_amd._moments import compute_moments
from cca_zoo_amd._utils._param_constraints import RIDGE_PARAMETER
from cca_zoo_amd._utils._validation import perview_parameter


class rCCA(BaseModel):
    """Regularised CCA (canonical ridge) for exactly two views.

    Args:
        latent_dimensions: number of canonical directions (default 1).
        center: subtract column means (stored in ``means_``) before fitting.
        c: ridge parameter(s) in ``[0, 1]``; scalar or ``[c1, c2]``.
    """

    _parameter_constraints: ClassVar[dict[str, list[Any]]] = {
        **BaseModel._parameter_constraints,
        "c": RIDGE_PARAMETER,
    }

    def __init__(self, latent_dimensions: int = 1, center: bool = True, c: float | list[float] = 0.0) -> None:
        super().__init__(latent_dimensions=latent_dimensions, center=center)
        self.c = c

    def _check_n_views(self) -> None:
        if self.n_views_ != 2:
            raise ValueError(
                f"rCCA requires exactly 2 views, got {self.n_views_}. "
                "Use MCCA for more than 2 views."
            )

    def _fit_moments(self, h, mom, n_total, dims, kind) -> None:
        c_ = perview_parameter("c", self.c, 0.0, 2)
        W, means, vals = h.rcca_solve(mom, n_total, dims, c_, self.center, self.latent_dimensions)
        self._store(W, means, kind, weights_like_input=True)
        self.singular_values_ = vals

    def fit(self, views, y=None):
        views_ = self._setup_fit(views)
        self._check_n_views()
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


class CCA(rCCA):
    """Canonical Correlation Analysis: :class:`rCCA` with ``c = 0``."""

    def __init__(self, latent_dimensions: int = 1, center: bool = True) -> None:
        super().__init__(latent_dimensions=latent_dimensions, center=center, c=0.0)

    def fit(self, views, y=None):
        return super().fit(views, y)


class PLS(rCCA):
    """Partial Least Squares (covariance maximisation): :class:`rCCA` with ``c = 1``."""

    def __init__(self, latent_dimensions: int = 1, center: bool = True) -> None:
        super().__init__(latent_dimensions=latent_dimensions, center=center, c=1.0)

    def fit(self, views, y=None):
        return super().fit(views, y)
