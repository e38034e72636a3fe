"""View validation -- the boundary semantics of cca_zoo/_utils/_validation.py:14-75.

Besides array-likes (validated with sklearn ``check_array(dtype="numeric")``, which
preserves float32 exactly like the reference), views may be **torch CUDA tensors**:
those stay in HBM and are handed to libccz by pointer (no host round trip).
"""

from __future__ import annotations

from typing import Any

import numpy as np
from sklearn.utils.validation import check_array


def is_device_tensor(v: Any) -> bool:
    return type(v).__module__.startswith("torch") and hasattr(v, "is_cuda") and bool(v.is_cuda)


def _n_rows(v) -> int:
    return int(v.shape[0])


def validate_views(views, min_views: int = 2, check_finite: bool = True) -> list:
    """Return the views as 2-D numpy arrays (or untouched 2-D CUDA tensors).

    ``check_finite=False`` skips sklearn's host-side NaN/inf scan (a single-threaded pass over the
    whole input, 0.2 s per 4 GB): ``fit`` detects non-finite inputs for free from the column sums of
    the moments pass and ``transform`` from its n x k output, raising the same ``ValueError``.

    Raises:
        ValueError: fewer than ``min_views`` views, or unequal numbers of samples.
    """
    if len(views) < min_views:
        raise ValueError(f"At least {min_views} views are required, got {len(views)}.")
    out = []
    for v in views:
        if is_device_tensor(v):
            if v.dim() != 2:
                raise ValueError(f"Expected 2D tensor, got {v.dim()}D tensor instead.")
            if not v.dtype.is_floating_point or v.element_size() not in (4, 8):
                raise ValueError("device views must be float32 or float64 tensors")
            out.append(v)
        else:
            if type(v).__module__.startswith("torch"):
                v = v.detach().cpu().numpy()
            out.append(check_array(v, ensure_2d=True, allow_nd=False, dtype="numeric", ensure_all_finite=check_finite))
    n = _n_rows(out[0])
    if not all(_n_rows(v) == n for v in out):
        raise ValueError(
            "All views must have the same number of samples. "
            f"Got shapes: {[tuple(v.shape) for v in out]}."
        )
    return out


def perview_parameter(name: str, value, default, n_views: int) -> list:
    """One value per view: a list is taken as is (its length must be ``n_views``), a scalar is repeated and
    ``None`` stands for ``default`` (semantics and message of cca_zoo/_utils/_validation.py:45-75)."""
    if not isinstance(value, list):
        fill = default if value is None else value
        return [fill for _ in range(n_views)]
    if len(value) == n_views:
        return value
    raise ValueError(f"Parameter '{name}' must be a scalar or a list of length {n_views}, got length {len(value)}.")
