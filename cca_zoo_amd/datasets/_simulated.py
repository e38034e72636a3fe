"""Linear latent-variable multiview generator.

Host generator: bit-identical to the reference's ``JointData`` for a given seed
(cca_zoo/datasets/_simulated.py:49-130 -- one ``default_rng``; loadings drawn at
construction in view order; every ``sample()`` draws ``z`` then one noise block per
view).  ``sample_device`` draws the same *model* straight into HBM with libccz's
counter-based generator for the at-scale measurement inputs (32.8 GB at the
north-star shape cannot be generated on the host); ``oracle/rng.py`` restates it
in NumPy, so any row range can be regenerated on the host.
"""

from __future__ import annotations

import numpy as np


class JointData:
    """``x_i = Z W_i' + noise_i / sqrt(snr_i)`` with ``Z ~ N(0, I)``, ``W_i ~ N(0, 1)``.

    Args:
        n_views, n_samples, latent_dimensions, n_features, signal_to_noise, random_state:
            as in the reference.
        latent_scales: optional per-latent standard deviations (e.g. ``linspace(2, .5, k)``)
            giving a separated canonical spectrum; ``None`` keeps the reference's isotropic
            latents.  Applies to both ``sample`` and ``sample_device``.
    """

    def __init__(self, n_views: int = 2, n_samples: int = 100, latent_dimensions: int = 1,
                 n_features: int | list[int] = 10, signal_to_noise: float | list[float] = 1.0,
                 random_state: int | None = None, latent_scales=None) -> None:
        self.n_views = n_views
        self.n_samples = n_samples
        self.latent_dimensions = latent_dimensions
        self.n_features = n_features
        self.signal_to_noise = signal_to_noise
        self.random_state = random_state
        self.latent_scales = latent_scales
        self._rng = np.random.default_rng(random_state)
        self._features_per_view = self._broadcast_param(n_features, n_views, "n_features")
        self._snr_per_view = self._broadcast_param(signal_to_noise, n_views, "signal_to_noise")
        self._weights = [self._rng.standard_normal((p, latent_dimensions)) for p in self._features_per_view]

    @staticmethod
    def _broadcast_param(value, n_views, name):
        if isinstance(value, list):
            if len(value) != n_views:
                raise ValueError(
                    f"Parameter '{name}' must be a scalar or a list of length {n_views}, got {len(value)}."
                )
            return list(value)
        return [value] * n_views

    def sample(self) -> list[np.ndarray]:
        z = self._rng.standard_normal((self.n_samples, self.latent_dimensions))
        if self.latent_scales is not None:
            z = z * np.asarray(self.latent_scales, dtype=np.float64)
        views = []
        for w, snr in zip(self._weights, self._snr_per_view):
            signal = z @ w.T
            noise_std = 1.0 / np.sqrt(snr) if snr > 0 else 1.0
            views.append(signal + self._rng.standard_normal(signal.shape) * noise_std)
        return views

    def __call__(self) -> list[np.ndarray]:
        return self.sample()

    def sample_device(self, device="cuda", dtype=None, n_samples=None, seed=None, *, row0=0, row_chunk=131072):
        """Draw rows ``[row0, row0 + n_samples)`` of the views directly into HBM (torch CUDA tensors).

        Every element is a pure function of ``(seed, global row, column)`` (``ccz_randn_fill``: SplitMix64 +
        Box-Muller, ``csrc/rng_hash.h``), the signal ``z (W diag(scales))'`` is a device GEMM (``ccz_transform``):
        ``x_v[r, :] = fl(z[r, :]) Wt_v + N_v[r, :] / sqrt(snr_v)``.  The same rows come out whatever the chunking,
        the shard (``row0``) or the number of GPUs, and ``oracle.rng.joint_data_rows`` regenerates any of them on
        the host."""
        import ctypes as C

        import torch

        from cca_zoo_amd import _backend

        dtype = dtype or torch.float32
        if dtype not in (torch.float32, torch.float64):
            raise TypeError("sample_device: dtype must be torch.float32 or torch.float64")
        n = int(n_samples if n_samples is not None else self.n_samples)
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("sample_device draws into HBM: a CUDA (ROCm) device is required")
        seed = int(self.random_state or 0) if seed is None else int(seed)
        base = (seed * 1000003) & 0xFFFFFFFFFFFFFFFF
        zseed = (base + 1) & 0xFFFFFFFFFFFFFFFF
        k = int(self.latent_dimensions)
        code = _backend.F32 if dtype == torch.float32 else _backend.F64
        outs = [torch.empty((n, p), dtype=dtype, device=dev) for p in self._features_per_view]
        h = _backend.handle_for(outs)
        scales = np.ones(k) if self.latent_scales is None else np.asarray(self.latent_scales, dtype=np.float64)
        wts = [torch.as_tensor(np.ascontiguousarray((w * scales[None, :]).T), dtype=torch.float64, device=dev)
               for w in self._weights]                                    # k x d_v
        zbuf = torch.empty((min(n, row_chunk), k), dtype=dtype, device=dev)
        sp = int(torch.cuda.current_stream(dev).cuda_stream)
        h.acquire(sp)                                      # every call below only enqueues work on libccz's stream
        for r0 in range(0, n, row_chunk):
            rows = min(n, r0 + row_chunk) - r0
            h.check(h.lib.ccz_randn_fill(h.raw, code, C.c_void_p(zbuf.data_ptr()), rows, k, k, zseed, int(row0) + r0,
                                         k + (k & 1), 1.0, 0))
            for v, (out, wt, snr) in enumerate(zip(outs, wts, self._snr_per_view)):
                d = int(out.shape[1])
                sd = 1.0 / float(np.sqrt(snr)) if snr > 0 else 1.0
                blk = out[r0:r0 + rows]
                h.check(h.lib.ccz_transform(h.raw, code, C.c_void_p(zbuf.data_ptr()), rows, k, k, None,
                                            C.c_void_p(wt.data_ptr()), d, C.c_void_p(blk.data_ptr()), d))
                h.check(h.lib.ccz_randn_fill(h.raw, code, C.c_void_p(blk.data_ptr()), rows, d, d,
                                             (base + 2 + v) & 0xFFFFFFFFFFFFFFFF, int(row0) + r0, d + (d & 1), sd, 1))
        h.release(sp)
        h.sync()                                           # one wait per draw: the views are complete when this returns
        return outs
