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



#: bench.py: record (start, head done, tail unpacked) events around the two parts of the exchange -- no host waits added
TIME_EXCHANGE = False
_last_events = None


def exchange_events():
    """The event triple of the last sharded ``compute_moments`` on CUDA (``None`` otherwise); read it after a
    synchronisation: ``start.elapsed_time(head)``, ``head.elapsed_time(tail)``."""
    return _last_events


def _exchange_buffer(h, dev, count):
    """The packed moments travel in ONE buffer per HANDLE and size, kept for the life of the handle: no second D^2-sized
    allocation per fit (it is 1 GB at D = 16384).  Keyed by the handle (a handle is single-threaded): two fits of the same
    size on one device from different handles / threads never share a payload."""
    import torch

    store = h.__dict__.setdefault("_exchange", {})
    buf = store.get(int(count))
    if buf is None:
        for k in [k for k in store if isinstance(k, int)]:
            del store[k]                                        # a different problem size: let the old buffer go
        buf = store[int(count)] = torch.empty(int(count), dtype=torch.float64, device=dev)
    return buf


def _side_stream(h, dev):
    import torch

    store = h.__dict__.setdefault("_exchange", {})
    st = store.get("stream")
    if st is None:
        st = store["stream"] = torch.cuda.Stream(device=dev)
    return st


def settle_deferred(h):
    """Error path of a fit that called ``compute_moments(defer_offdiag=True)``: the deferred half of the exchange may still be
    writing into the moments buffer on another stream.  Wait for it on the device (``ccz_solve_defer(NULL)`` also clears the
    registration) and then for the handle, so that the buffer can be dropped safely."""
    try:
        h.solve_defer(None)
        h.sync()
    except Exception:
        pass


def compute_moments(views, handle=None, defer_offdiag=False):
    """``defer_offdiag`` (the estimators' ``fit`` inside ``row_sharded()``): the exchange is issued in two parts --
    diagonal blocks + column sums + row count first, the off-diagonal blocks second -- and this function returns as soon
    as the FIRST part has arrived; the second is unpacked on a side stream and ``ccz_solve_defer`` makes the next
    ``ccz_*_solve`` wait for it on the device right before it reads an off-diagonal block, i.e. after the Cholesky
    chain of the diagonal blocks.  Consumers that read the moments themselves must leave it off."""
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
    ccz_comm = _dist.active_group() if sharded and isinstance(_dist.active_group(), _dist.CczComm) else None
    stream_ptr = 0
    if on_device or (sharded and ccz_comm is None):
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
    if ccz_comm is not None:
        # the exchange behind the C ABI (no torch.distributed), ONE call: the blocks layout in a buffer the handle keeps, the
        # row count written on the device, head and tail reduced on the handle's exchange stream, the tail still in flight when
        # this returns -- the solve waits for it on the device right before its first off-diagonal read (ccz_moments_exchange)
        if ccz_comm.handle is not h:
            raise ValueError("the CczComm of row_sharded() belongs to another handle / device than the views")
        if on_device:
            h.acquire(stream_ptr)
        t_ar = time.perf_counter()
        n_total = h.moments_exchange(mom_ptr, D, dims, n)
        LAST["allreduce_ms"] = (time.perf_counter() - t_ar) * 1e3
        if not defer_offdiag:
            # the caller is not a solve (score, grid search, partial / group estimators): it reads off-diagonal blocks through
            # other entry points, which do not wait for the tail -- consume the deferral now (a device-side wait)
            h.solve_defer(None)
    elif sharded:
        # the one exchange step of the path, in two parts: [diag-block triangles | column sums | row count] and
        # [off-diagonal blocks] (ccz.h "blocks layout") -- D (D + 1) / 2 + D + 1 doubles in all, as the plain packed form
        import torch
        import torch.distributed as dist

        group = _dist.active_group()
        n_head = sum(d * (d + 1) // 2 for d in dims) + D + 1
        n_tail = D * (D + 1) // 2 + D + 1 - n_head
        cuda = mom_t.is_cuda
        packed = _exchange_buffer(h, mom_t.device, n_head + n_tail) if cuda else torch.empty(n_head + n_tail, dtype=torch.float64)
        h.moments_pack_blocks(mom_ptr, D, dims, packed.data_ptr(), h.BOTH)
        if cuda:
            h.release(stream_ptr)                        # the collective's stream follows libccz's on the device
        else:
            h.sync()
        head, tail = packed[:n_head], packed[n_head:]
        head[-1:].fill_(float(n))                        # the row count rides in the last slot of the head
        global _last_events
        _last_events = None
        timed = TIME_EXCHANGE and cuda
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        t_ar = time.perf_counter()
        staged = cuda and _dist.staged_over_gloo(packed, group)
        if staged:
            # test transport (two ranks on ONE GPU, _dist.staged_over_gloo): the same two-part order with the collectives on
            # pinned host copies; the tail's result goes back on the side stream and is unpacked there behind the deferral
            # event exactly as on the nccl route (the host does block for the tail's collective here)
            host = torch.empty(n_head + n_tail, dtype=torch.float64, pin_memory=True)
            host.copy_(packed)
            dist.all_reduce(host[:n_head], op=dist.ReduceOp.SUM, group=group)
            head.copy_(host[:n_head], non_blocking=True)
            keep.append(host)

            class _Tail:
                def wait(self_inner):
                    dist.all_reduce(host[n_head:], op=dist.ReduceOp.SUM, group=group)
                    tail.copy_(host[n_head:], non_blocking=True)     # on the stream that is current when waited for

            w_head, w_tail = None, (_Tail() if n_tail > 0 else None)
        else:
            w_head = dist.all_reduce(head, op=dist.ReduceOp.SUM, group=group, async_op=True)
            w_tail = dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=group, async_op=True) if n_tail > 0 else None
            w_head.wait()
        if timed:
            ev1.record()
            _last_events = (ev0, ev1, None)
        n_total = int(round(float(head[-1].item())))     # the one host synchronisation: n is a host argument of the solves
        LAST["allreduce_ms"] = (time.perf_counter() - t_ar) * 1e3
        if cuda:
            h.acquire(stream_ptr)
        h.moments_unpack_blocks(packed.data_ptr(), D, dims, mom_ptr, h.HEAD)
        if w_tail is not None:
            if cuda and defer_offdiag:
                side = _side_stream(h, mom_t.device)
                side.wait_stream(torch.cuda.current_stream(mom_t.device))
                with torch.cuda.stream(side):
                    w_tail.wait()                        # the side stream waits for the collective, the host does not
                # libccz records an event it OWNS behind this unpack and the next *_solve waits for it on the device
                # (every solve entry consumes the registration on every exit path); the side stream's writes into
                # mom_t are made known to torch's allocator
                h.moments_unpack_blocks(packed.data_ptr(), D, dims, mom_ptr, h.TAIL, on_stream=side.cuda_stream)
                mom_t.record_stream(side)
                if timed:
                    ev2 = torch.cuda.Event(enable_timing=True)
                    ev2.record(side)
                    _last_events = (ev0, ev1, ev2)
            else:
                w_tail.wait()
                if cuda:
                    h.acquire(stream_ptr)
                h.moments_unpack_blocks(packed.data_ptr(), D, dims, mom_ptr, h.TAIL)
        keep.append(packed)
    # non-finite inputs (NaN / inf anywhere in a column) surface in that column's sum: the reference's
    # check_array(force_all_finite) ValueError without a host pass over the data
    sums = h.to_host(mom_ptr, (D,), offset_bytes=D * D * 8)
    if not np.all(np.isfinite(sums)):
        if sharded and defer_offdiag:
            settle_deferred(h)                           # the tail may still be in flight into mom: no solve will consume it
        bad = int(np.flatnonzero(~np.isfinite(sums))[0])
        raise ValueError(f"Input contains NaN or infinity (first affected stacked column: {bad}).")
    # no symmetrisation pass: the solvers read the upper triangle (authoritative) on both sides
    return mom_ptr, keep, n_total, dims, kind
