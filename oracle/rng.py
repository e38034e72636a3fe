"""NumPy restatement of the counter-based normal generator of the at-scale inputs.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  ``cca_zoo_amd/csrc/rng_hash.h::hash_normal_pair`` is a pure
function of ``(seed, pair index)``: SplitMix64 hashes -> two uniforms -> Box-Muller (cos for the even element of the
pair, sin for the odd one).  ``randn_block`` reproduces ``ccz_randn_fill`` for any row range, and ``joint_data_rows``
the views that ``cca_zoo_amd.datasets.JointData.sample_device`` writes into HBM (the latent-variable model of
cca_zoo/datasets/_simulated.py:116-130), so a parity test can regenerate a slice of the 32.8 GB bench inputs on the
host.  Agreement with the device is to the accuracy of ``log`` / ``cos`` / ``sin`` (a few ulp of float64), not bitwise.
"""

from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def normal_pairs(seed: int, pair):
    """(n0, n1) of the Box-Muller pairs ``pair`` (uint64 array) of stream ``seed``."""
    pair = np.asarray(pair, dtype=np.uint64)
    s = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        a = splitmix64(s ^ splitmix64(np.uint64(2) * pair))
        b = splitmix64(s ^ splitmix64(np.uint64(2) * pair + np.uint64(1)))
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)
    u2 = (b >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    r = np.sqrt(-2.0 * np.log(u1))
    t = 6.283185307179586 * u2
    return r * np.cos(t), r * np.sin(t)


def randn_block(seed: int, row0: int, rows: int, cols: int, row_stride: int | None = None) -> np.ndarray:
    """``out[r, c] = N(seed, (row0 + r) * row_stride + c)`` as ``ccz_randn_fill`` defines it (float64)."""
    row_stride = int(row_stride if row_stride is not None else cols + (cols & 1))
    if row_stride < cols or row_stride & 1:
        raise ValueError("row_stride must be even and >= cols")
    half = (cols + 1) // 2
    r = np.arange(row0, row0 + rows, dtype=np.uint64)[:, None]
    q = np.arange(half, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        pair = (r * np.uint64(row_stride)) // np.uint64(2) + q
    n0, n1 = normal_pairs(seed, pair)
    out = np.empty((rows, 2 * half))
    out[:, 0::2] = n0
    out[:, 1::2] = n1
    return out[:, :cols]


def stream_seeds(seed: int, n_views: int):
    """Seeds of the latent stream and of the per-view noise streams (same rule as ``JointData.sample_device``)."""
    base = (int(seed) * 1000003) & 0xFFFFFFFFFFFFFFFF
    return (base + 1) & 0xFFFFFFFFFFFFFFFF, [(base + 2 + v) & 0xFFFFFFFFFFFFFFFF for v in range(n_views)]


def joint_data_rows(weights, snr, latent_scales, seed: int, row0: int, rows: int, dtype=np.float32):
    """Rows ``[row0, row0 + rows)`` of the views ``JointData.sample_device(seed=seed)`` draws:
    ``x_v = fl(z) (W_v diag(scales))' + N_v / sqrt(snr_v)`` with z rounded to the view dtype first (the device
    forms the signal with a GEMM in that dtype) and the result rounded to ``dtype``."""
    k = int(weights[0].shape[1])
    zseed, vseeds = stream_seeds(seed, len(weights))
    z = randn_block(zseed, row0, rows, k).astype(dtype).astype(np.float64)
    sc = np.ones(k) if latent_scales is None else np.asarray(latent_scales, dtype=np.float64)
    out = []
    for w, s_n, vs in zip(weights, snr, vseeds):
        d = int(w.shape[0])
        sd = 1.0 / np.sqrt(s_n) if s_n > 0 else 1.0
        wt = (np.asarray(w, dtype=np.float64) * sc[None, :]).T            # k x d
        if np.dtype(dtype) == np.float32:
            wt = wt.astype(np.float32).astype(np.float64)                 # the device converts W to fp32 for the fp32 GEMM
        signal = (z @ wt).astype(dtype).astype(np.float64)
        out.append((signal + sd * randn_block(vs, row0, rows, d)).astype(dtype))
    return out
