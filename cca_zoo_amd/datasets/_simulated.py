"""Linear latent-variable multiview generator.

Host generator: bit-identical to the reference's ``JointData`` for a given seed
(cca_zoo/datasets/_simulated.py:49-130 -- one ``default_rng``; loadings drawn at
construction in view order; every ``sample()`` draws ``z`` then one noise block per
view).  ``sample_device`` draws the same *model* straight into HBM with torch's
Philox generator for the at-scale measurement inputs (32.8 GB at the north-star
shape cannot be generated on the host); its streams are torch's, not NumPy's.
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

    def sample_device(self, device="cuda", dtype=None, n_samples=None, seed=None, row_chunk=65536):
        """Draw the views directly into HBM (torch CUDA tensors), chunked over rows."""
        import torch

        dtype = dtype or torch.float32
        n = int(n_samples if n_samples is not None else self.n_samples)
        gen = torch.Generator(device=device)
        gen.manual_seed(int(self.random_state or 0) if seed is None else int(seed))
        Ws = [torch.as_tensor(w, dtype=torch.float32, device=device) for w in self._weights]
        scales = None
        if self.latent_scales is not None:
            scales = torch.as_tensor(np.asarray(self.latent_scales), dtype=torch.float32, device=device)
        outs = [torch.empty((n, p), dtype=dtype, device=device) for p in self._features_per_view]
        for r0 in range(0, n, row_chunk):
            r1 = min(n, r0 + row_chunk)
            z = torch.randn((r1 - r0, self.latent_dimensions), generator=gen, device=device, dtype=torch.float32)
            if scales is not None:
                z = z * scales
            for out, w, snr in zip(outs, Ws, self._snr_per_view):
                sd = 1.0 / float(np.sqrt(snr)) if snr > 0 else 1.0
                blk = torch.randn((r1 - r0, w.shape[0]), generator=gen, device=device, dtype=torch.float32)
                blk.mul_(sd).addmm_(z, w.T)
                out[r0:r1].copy_(blk)
        return outs
