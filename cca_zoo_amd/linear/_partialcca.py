"""Partial CCA from second moments (reference: cca_zoo/linear/_partialcca.py:14-176).

The reference regresses the confounds ``Z`` out of every (centred) view with ``pinv(Z)`` -- an n x dz
SVD plus two n x d x dz products per view -- and runs the MCCA generalised eigenproblem (``_build_A`` /
``_build_B`` hooks, ``pca=False``) on the n x d residuals.  Everything it needs is a function of the
second moments of ``[Z | X_1 .. X_m]``:

    Xc = X - 1 mu'                     (mu = column means if ``center`` else 0)
    Z'Xc = G_zx - s_z mu'              beta = pinv(Z) Xc = (Z'Z)^+ Z'Xc
    R = Xc - Z beta                    R'R  = Xc'Xc - (Z'Xc)' (Z'Z)^+ (Z'Xc),   1'R = 1'Xc - s_z' beta

so ONE K1 pass over ``[Z | views]`` is followed by a dz x dz pseudo-inverse on the host, one
rank-(1 + dz) GEMM on the device that turns the views' block of G into ``R'R`` (``ccz_moments_subset`` +
``ccz_gemm_f64``), and the ordinary ``ccz_mcca_solve`` on these *effective moments* (its covariance
``(R'R - (1'R)'(1'R)/n)/(n-1)`` is exactly the ``np.cov`` of the residuals the reference builds).
"""

from __future__ import annotations

import numpy as np
from sklearn.utils.validation import check_is_fitted

from cca_zoo_amd import _backend
from cca_zoo_amd._moments import compute_moments
from cca_zoo_amd._utils._validation import is_device_tensor, perview_parameter, validate_views
from cca_zoo_amd.linear._mcca import MCCA


def _as_partials(partials, like):
    """Confounds as a 2-D array of the same residency as the views (host ndarray / CUDA tensor)."""
    if is_device_tensor(like):
        import torch

        p = partials if is_device_tensor(partials) else torch.as_tensor(np.asarray(partials, dtype=float), device=like.device)
        p = p.to(like.dtype) if p.dtype != like.dtype else p
        return p.reshape(p.shape[0], -1)
    p = partials.detach().cpu().numpy() if type(partials).__module__.startswith("torch") else np.asarray(partials, dtype=float)
    return p.reshape(p.shape[0], -1)


class PartialCCA(MCCA):
    """CCA of the views after regressing out confounding variables ``partials``.

    Args:
        latent_dimensions, center, c, eps: as :class:`MCCA` (``pca`` is fixed to ``False``).
    """

    def __init__(self, latent_dimensions: int = 1, center: bool = True, c: float | list[float] = 0.0,
                 eps: float = 1e-6) -> None:
        super().__init__(latent_dimensions=latent_dimensions, center=center, c=c, pca=False, eps=eps)

    def fit(self, views, y=None, partials=None):
        if partials is None:
            raise ValueError("PartialCCA requires `partials` to be provided to fit().")
        views_ = self._setup_fit(views)
        Z = _as_partials(partials, views_[0])
        if int(Z.shape[0]) != self.n_samples_:
            raise ValueError("`partials` must have one row per sample")
        dz = int(Z.shape[1])
        h = _backend.handle_for(views_)
        # K1 over [Z | X_1 .. X_m]: the confound rows of G are the first dz rows (all inside the upper triangle)
        mom, keep, n, dims_all, kind = compute_moments([Z, *views_], h)
        self.n_samples_ = int(n)
        dims = dims_all[1:]
        D, Da = int(sum(dims)), int(sum(dims_all))
        top = h.to_host(mom, (dz, Da))                                   # [Z'Z | Z'X]
        s_all = h.to_host(mom, (Da,), offset_bytes=Da * Da * 8)
        Gzz, Gzx, s_z, s_x = top[:, :dz], top[:, dz:], s_all[:dz], s_all[dz:]
        mu = s_x / n if self.center else np.zeros(D)
        ZtXc = Gzx - np.outer(s_z, mu)
        Gzz = np.triu(Gzz) + np.triu(Gzz, 1).T                           # K1 fills upper-triangular tiles
        beta = np.linalg.pinv(Gzz, hermitian=True) @ ZtXc                # = pinv(Z) Xc   (dz x D)
        # effective moments of the residual views: G_eff = Xc'Xc - (Z'Xc)' beta,  s_eff = 1'Xc - s_z' beta
        eff = h.alloc((D * D + D) * 8)
        h.moments_subset(mom, Da, dz, D, eff.ptr)
        U = np.vstack([ZtXc, (s_x / np.sqrt(n))[None, :]]) if self.center else ZtXc
        V = np.vstack([beta, (s_x / np.sqrt(n))[None, :]]) if self.center else beta
        Ud, Vd = h.to_device(U), h.to_device(V)
        h.gemm(1, 0, D, D, U.shape[0], -1.0, Ud.ptr, D, Vd.ptr, D, 1.0, eff.ptr, D)
        s_eff = (s_x - n * mu) - s_z @ beta
        h.h2d(eff.ptr + D * D * 8, s_eff)
        c_ = perview_parameter("c", self.c, 0.0, self.n_views_)
        W, _, vals = h.mcca_solve(eff.ptr, n, dims, c_, self.eps, True, self.latent_dimensions)
        self._store(W, np.split(mu, np.cumsum(dims)[:-1]), kind, weights_like_input=False)
        self.eigenvalues_ = vals
        self.confound_betas_ = [np.ascontiguousarray(b) for b in np.split(beta, np.cumsum(dims)[:-1], axis=1)]
        del keep
        return self

    def transform(self, views, partials=None):
        """``(X_i - mean_i - Z beta_i) W_i``; without ``partials`` the plain projection (as the reference)."""
        check_is_fitted(self)
        base = super().transform(views)
        if partials is None:
            return base
        validated = validate_views(views, check_finite=False)
        Z = _as_partials(partials, validated[0])
        out = []
        for z, b, w in zip(base, self.confound_betas_, self.weights_):
            bw = b @ w                                                   # dz x k  (host, tiny)
            if is_device_tensor(z):
                from cca_zoo_amd._base import _device_project

                # Z (beta W) on the device through ccz_transform (no vendor BLAS on the product path)
                out.append(z - _device_project(Z.to(z.dtype), np.zeros(bw.shape[0]), bw))
            else:
                from cca_zoo_amd._base import _host_project

                out.append(z - _host_project(np.ascontiguousarray(Z, dtype=z.dtype), np.zeros(bw.shape[0]), bw))
        return out

    def fit_transform(self, views, y=None, partials=None):
        return self.fit(views, y=y, partials=partials).transform(views, partials=partials)
