"""Host-side driver of K1: second moments of a list of views (+ the all-reduce).

``views`` are what ``validate_views`` returned: C-contiguous-izable numpy arrays
(streamed to the device in row chunks by libccz) or torch CUDA tensors (handed over
by pointer).  Returns ``(moments_ptr, keepalive, n_total, dims, in_dtype)`` with the
moments (upper-triangular tiles of G, column sums) and -- inside ``row_sharded()`` -- summed over all ranks.
"""

from __future__ import annotations

import threading
import time

import numpy as np

from cca_zoo_amd import _backend, _dist
from cca_zoo_amd._utils._validation import is_device_tensor


class _Last(threading.local):
    """Per-THREAD record (a handle is single-threaded; concurrent fits on different handles keep their own numbers)."""

    def __init__(self):
        self.values = {"moments_ms": 0.0, "allreduce_ms": 0.0}

    def __getitem__(self, key):
        return self.values[key]

    def __setitem__(self, key, value):
        self.values[key] = value


#: wall-clock of the pieces of the last ``compute_moments`` call on this thread (bench.py reports them)
LAST = _Last()


def _common_float(views):
    """float32 only if every view is float32; anything else computes in float64
    (the reference's NumPy promotion of mixed inputs ends in float64 too)."""
    kinds = []
    for v in views:
        if is_device_tensor(v):
            kinds.append("f32" if v.element_size() == 4 else "f64")
        else:
            kinds.append("f32" if v.dtype == np.float32 else "f64")
    return "f32" if all(k == "f32" for k in kinds) else "f64"


def compute_moments(views, handle=None):
    h = handle or _backend.handle_for(views)
    n = int(views[0].shape[0])
    dims = [int(v.shape[1]) for v in views]
    D = sum(dims)
    kind = _common_float(views)
    on_device = all(is_device_tensor(v) for v in views)
    if any(is_device_tensor(v) for v in views) and not on_device:
        raise ValueError("views must be all host arrays or all CUDA tensors")
    keep = []
    sharded = _dist.is_sharded()
    if on_device or sharded:
        import torch

        # (a test double of the library keeps "device" memory on the host and says so: gloo tests on CPU)
        dev = views[0].device if on_device else (getattr(h, "torch_device", None) or torch.device("cuda", h.device))
        mom_t = torch.empty(D * D + D, dtype=torch.float64, device=dev)
        mom_ptr = mom_t.data_ptr()
        keep.append(mom_t)
    else:
        mom_buf = h.alloc((D * D + D) * 8)
        mom_ptr = mom_buf.ptr
        keep.append(mom_buf)
        mom_t = None
    descr = []
    if on_device:
        import torch

        tdt = torch.float32 if kind == "f32" else torch.float64
        for v in views:
            if v.dtype != tdt or v.stride(1) != 1 or v.stride(0) < v.shape[1]:     # (expanded / overlapping rows: ld >= d)
                v = v.to(tdt).contiguous()
            keep.append(v)
            descr.append((v.data_ptr(), v.shape[1], v.stride(0)))
        stream_ptr = int(torch.cuda.current_stream(views[0].device).cuda_stream)
        h.acquire(stream_ptr)                              # libccz's stream follows torch's on the device; no host wait
    else:
        ndt = np.float32 if kind == "f32" else np.float64
        for v in views:
            a = np.ascontiguousarray(v, dtype=ndt)
            keep.append(a)
            descr.append((a, a.shape[1], a.shape[1]))
    t_k1 = time.perf_counter()
    h.moments(descr, n, _backend.F32 if kind == "f32" else _backend.F64, on_device, mom_ptr, accumulate=False)
    LAST["moments_ms"] = (time.perf_counter() - t_k1) * 1e3
    LAST["allreduce_ms"] = 0.0
    n_total = n
    if sharded:
        # the one collective of the path: packed upper triangle + column sums + row count
        import torch

        npk = D * (D + 1) // 2 + D
        packed = torch.empty(npk + 1, dtype=torch.float64, device=mom_t.device)
        h.moments_pack(mom_ptr, D, packed.data_ptr())
        if on_device:
            h.release(stream_ptr)                        # the collective's stream follows libccz's on the device
        else:
            h.sync()
        packed[npk:].fill_(float(n))                     # the row count rides in the tail slot of the same buffer
        t_ar = time.perf_counter()
        n_total = _dist.allreduce_moments(packed, _dist.active_group())     # in place; its .item() is the sync
        LAST["allreduce_ms"] = (time.perf_counter() - t_ar) * 1e3
        h.moments_unpack(packed.data_ptr(), D, mom_ptr)
        keep.append(packed)
    # non-finite inputs (NaN / inf anywhere in a column) surface in that column's sum: the reference's
    # check_array(force_all_finite) ValueError without a host pass over the data
    sums = h.to_host(mom_ptr, (D,), offset_bytes=D * D * 8)
    if not np.all(np.isfinite(sums)):
        bad = int(np.flatnonzero(~np.isfinite(sums))[0])
        raise ValueError(f"Input contains NaN or infinity (first affected stacked column: {bad}).")
    # no symmetrisation pass: the solvers read the upper triangle (authoritative) on both sides
    return mom_ptr, keep, n_total, dims, kind
