"""Estimator surface shared by the device-backed CCA models.

Same public contract as the reference's ``BaseModel`` (cca_zoo/_base.py:19-258):
sklearn ``BaseEstimator`` (``get_params/set_params/clone/repr``), constructor
validation via ``_parameter_constraints``, ``fit`` sets ``weights_`` / ``means_`` /
``n_views_`` / ``n_features_in_`` / ``n_samples_``; ``transform``, ``fit_transform``,
``score``, ``pairwise_correlations``, ``average_pairwise_correlations``, ``weights`` and
``get_factor_loadings`` derive from those.  What differs is underneath: ``fit`` never
materialises centred copies -- the column sums come out of the same pass that builds
the Gram matrix (libccz K1) and centring is applied to the d x d moments.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from numbers import Integral
from typing import Any, ClassVar

import numpy as np
from sklearn.base import BaseEstimator
from sklearn.utils import Tags
from sklearn.utils._param_validation import Interval
from sklearn.utils.validation import check_is_fitted

from cca_zoo_amd._utils._validation import is_device_tensor, validate_views


class BaseModel(BaseEstimator, ABC):
    """Abstract base of the multiview estimators (see module docstring)."""

    _parameter_constraints: ClassVar[dict[str, list[Any]]] = {
        "latent_dimensions": [Interval(Integral, 1, None, closed="left")],
        "center": ["boolean"],
    }

    def __init__(self, latent_dimensions: int = 1, center: bool = True) -> None:
        self.latent_dimensions = latent_dimensions
        self.center = center

    @abstractmethod
    def fit(self, views, y=None):
        """Fit to a list of (n_samples, n_features_i) views; returns ``self``."""

    # -- fit plumbing ------------------------------------------------------------
    def _setup_fit(self, views) -> list:
        """Validate parameters + views and record the metadata of cca_zoo/_base.py:92-96.

        Unlike the reference this does NOT return centred copies: ``means_`` is filled in
        by the concrete ``fit`` from the column sums of the moments pass.
        """
        self._validate_params()
        validated = validate_views(views, check_finite=False)   # checked on the moments (compute_moments)
        self.n_views_ = len(validated)
        self.n_features_in_ = [int(v.shape[1]) for v in validated]
        self.n_samples_ = int(validated[0].shape[0])
        return validated

    def _check_n_views(self) -> None:
        """Hook: estimators restricted to a fixed number of views raise here."""

    def _fit_moments(self, h, mom, n_total, dims, kind) -> None:
        """Solve stage: fill ``weights_`` / ``means_`` from the device moments ``[G | s]`` of ``n_total`` rows."""
        raise NotImplementedError

    def _fit_from_moments(self, h, mom, n_total, dims, kind):
        """``fit`` without a pass over the data: everything after K1.  Used by
        :class:`cca_zoo_amd.model_selection.GridSearchCV`, which reuses one set of moments for every
        hyper-parameter setting and obtains the training moments of a fold by subtraction."""
        self._validate_params()
        self.n_views_ = len(dims)
        self.n_features_in_ = [int(d) for d in dims]
        self.n_samples_ = int(n_total)
        self._check_n_views()
        self._fit_moments(h, mom, n_total, dims, kind)
        return self

    def _store(self, weights, means, in_kind, weights_like_input):
        """dtype flow of the reference: rCCA keeps the input dtype for weights; MCCA/GCCA
        promote to float64 (np.cov); ``means_`` follow the input dtype when centring, else
        float64 zeros (np.zeros(p), _base.py:101)."""
        wdt = np.float32 if (in_kind == "f32" and weights_like_input) else np.float64
        mdt = np.float32 if (in_kind == "f32" and self.center) else np.float64
        self.weights_ = [np.ascontiguousarray(w, dtype=wdt) for w in weights]
        self.means_ = [np.ascontiguousarray(m, dtype=mdt) for m in means]

    # -- public API ---------------------------------------------------------------------
    def transform(self, views) -> list:
        """``(X_i - mean_i) @ W_i`` per view.  Host arrays in -> numpy out (device GEMM inside
        libccz); CUDA tensors in -> CUDA tensors out."""
        check_is_fitted(self)
        validated = validate_views(views, check_finite=False)
        out = []
        for v, m, w in zip(validated, self.means_, self.weights_):
            if is_device_tensor(v):
                out.append(_device_project(v, m, w))
            else:
                z = _host_project(v, m, w)
                if not np.all(np.isfinite(z)):   # NaN / inf rows of the input propagate to the n x k output
                    raise ValueError("Input contains NaN or infinity.")
                out.append(z)
        return out

    def fit_transform(self, views, y=None) -> list:
        return self.fit(views, y).transform(views)

    def score(self, views, y=None) -> np.ndarray:
        return self.average_pairwise_correlations(views)

    def pairwise_correlations(self, views) -> np.ndarray:
        """(n_views, n_views, k) Pearson correlations between canonical variates.

        The reference stacks the n x k variates and reduces them on the host (``_base.py:150-172``); here the
        variates (host arrays or CUDA tensors, as ``transform`` returned them) go through one K1 pass
        (``ccz_moments`` on the m*k stacked columns) and the correlations are read off their second moments:
        ``corr = (G_ab - s_a s_b / n) / (nrm_a nrm_b)``, ``nrm_a^2 = G_aa - s_a^2 / n``, with the reference's
        guard (``nrm <= 1e-12`` -> divide by 1).  Inside ``row_sharded()`` the moments are all-reduced, i.e. the
        correlations are those of the global sample."""
        from cca_zoo_amd import _backend
        from cca_zoo_amd._moments import compute_moments

        zs = self.transform(views)
        m, k = len(zs), int(zs[0].shape[1])
        h = _backend.handle_for(zs)
        mom, keep, n, _, _ = compute_moments(zs, h)
        D = m * k
        h.moments_symmetrize(mom, D)
        flat = h.to_host(mom, (D * D + D,))
        del keep
        G, s = flat[: D * D].reshape(D, D), flat[D * D:]
        S = G - np.outer(s, s) / n
        nrm = np.sqrt(np.maximum(np.diag(S), 0.0))
        nrm = np.where(nrm > 1e-12, nrm, 1.0)
        R = S / np.outer(nrm, nrm)
        idx = np.arange(k)
        out = np.empty((m, m, k))
        for i in range(m):
            for j in range(m):
                out[i, j] = R[i * k + idx, j * k + idx]
        return out

    def average_pairwise_correlations(self, views) -> np.ndarray:
        R = self.pairwise_correlations(views)
        m = R.shape[0]
        off = R.sum(axis=(0, 1)) - sum(R[i, i, :] for i in range(m))
        return off / (m * (m - 1))

    @property
    def weights(self) -> list:
        check_is_fitted(self)
        return self.weights_

    def get_factor_loadings(self, views) -> list:
        """Pearson correlation of every input feature with every canonical variate.

        The reference forms the n x k variates and an n x d x k product per view on the host
        (``_base.py:208-234``).  Both the covariance between features and variates and their variances are
        functions of the view's second moments: ``cov(x, z) = C W``, ``var z = diag(W'CW)``, ``var x = diag(C)``
        -- so each view takes ONE K1 pass (``ccz_moments``; host arrays are streamed, CUDA tensors stay in HBM)
        and a d x d x k product on the device (``ccz_factor_loadings``).  Inside ``row_sharded()`` the moments are
        all-reduced, i.e. the loadings are those of the global sample."""
        import ctypes as C

        from cca_zoo_amd import _backend
        from cca_zoo_amd._moments import compute_moments

        check_is_fitted(self)
        validated = validate_views(views, check_finite=False)
        out = []
        for v, w in zip(validated, self.weights_):
            h = _backend.handle_for([v])
            mom, keep, n, dims, _ = compute_moments([v], h)
            d, k = int(dims[0]), int(w.shape[1])
            wd = h.to_device(np.ascontiguousarray(w, dtype=np.float64))
            od = h.alloc(d * k * 8)
            h.check(h.lib.ccz_factor_loadings(h.raw, C.c_void_p(int(mom)), int(n), d, C.c_void_p(wd.ptr), k,
                                              C.c_void_p(od.ptr)))
            out.append(h.to_host(od, (d, k)))
            del keep
        return out

    def __sklearn_tags__(self) -> Tags:
        tags = super().__sklearn_tags__()
        tags.no_validation = True
        tags.input_tags.two_d_array = False
        tags._skip_test = True
        return tags


def _host_project(v, mean, w):
    """(v - mean) @ w for a host array: staged through HBM, multiplied by ``ccz_transform``.
    Result dtype follows NumPy promotion of (v, mean, w) like the reference expression."""
    import ctypes as C

    from cca_zoo_amd import _backend

    rdt = np.result_type(v.dtype, mean.dtype, w.dtype)
    rdt = np.float32 if rdt == np.float32 else np.float64
    x = np.ascontiguousarray(v, dtype=rdt)
    n, d = x.shape
    k = int(w.shape[1])
    h = _backend.default_handle()
    xd = h.to_device(x)
    md = h.to_device(np.ascontiguousarray(mean, dtype=np.float64))
    wd = h.to_device(np.ascontiguousarray(w, dtype=np.float64))
    od = h.alloc(max(n * k * x.itemsize, 8))
    h.check(h.lib.ccz_transform(h.raw, _backend.F32 if rdt == np.float32 else _backend.F64,
                                C.c_void_p(xd.ptr), n, d, d, C.c_void_p(md.ptr), C.c_void_p(wd.ptr), k,
                                C.c_void_p(od.ptr), k))
    return h.to_host(od, (n, k), dtype=rdt)


def _device_project(v, mean, w):
    """(v - mean) @ w for a CUDA tensor through ``ccz_transform`` (HBM-resident)."""
    import ctypes as C

    import torch

    from cca_zoo_amd import _backend

    h = _backend.handle_for([v])
    if v.stride(1) != 1 or v.stride(0) < v.shape[1]:      # expanded / overlapping rows: ld must be >= d
        v = v.contiguous()
    k = int(w.shape[1])
    out = torch.empty((v.shape[0], k), dtype=v.dtype, device=v.device)
    md = torch.as_tensor(np.asarray(mean, dtype=np.float64), device=v.device)
    wd = torch.as_tensor(np.ascontiguousarray(w, dtype=np.float64), device=v.device)
    sp = int(torch.cuda.current_stream(v.device).cuda_stream)
    h.acquire(sp)                                          # device-side hand-over: the host never waits here
    h.check(h.lib.ccz_transform(h.raw, _backend.F32 if v.element_size() == 4 else _backend.F64,
                                C.c_void_p(v.data_ptr()), v.shape[0], v.shape[1], v.stride(0),
                                C.c_void_p(md.data_ptr()), C.c_void_p(wd.data_ptr()), k,
                                C.c_void_p(out.data_ptr()), out.stride(0)))
    h.release(sp)
    return out
