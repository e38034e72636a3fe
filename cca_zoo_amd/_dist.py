"""Sample-axis sharding: one process per GPU, one all-reduce of the moments.

The hot path shards naturally on rows: every rank computes the
second moments ``[G | s]`` of ITS rows with K1, a single ``all_reduce(SUM)`` over
RCCL/xGMI (``torch.distributed`` backend ``nccl``) makes them global, and the small
dense solves run replicated.  No other collective exists on the path.

Usage (under ``torchrun``)::

    with cca_zoo_amd.row_sharded():          # views passed to fit() are this rank's rows
        model.fit([X1_local, X2_local])

``shard_bounds`` gives the contiguous row partition used by bench.py and the tests.
"""

from __future__ import annotations

import contextlib
import threading

_state = threading.local()


def shard_bounds(n_rows: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced partition: the first ``n % world`` ranks get one extra row."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, extra = divmod(int(n_rows), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def active_group():
    """The process group of the enclosing ``row_sharded`` block, or ``None``."""
    return getattr(_state, "group", None)


def is_sharded() -> bool:
    return getattr(_state, "on", False)


@contextlib.contextmanager
def row_sharded(group=None):
    """Treat the views given to ``fit`` as this rank's row shard of a global dataset."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("row_sharded() needs an initialised torch.distributed process group")
    prev = (getattr(_state, "on", False), getattr(_state, "group", None))
    _state.on, _state.group = True, group
    try:
        yield
    finally:
        _state.on, _state.group = prev


@contextlib.contextmanager
def unsharded():
    """Inside ``row_sharded()``: treat the enclosed calls as purely local (no collective) -- e.g. a spot check of this
    rank's own rows."""
    prev = (getattr(_state, "on", False), getattr(_state, "group", None))
    _state.on, _state.group = False, None
    try:
        yield
    finally:
        _state.on, _state.group = prev


def allreduce_moments(buf, group=None):
    """SUM-reduce ``buf`` IN PLACE and return the global row count.

    ``buf`` is a flat float64 torch tensor ``[payload ... | n_local]`` (CUDA for nccl, CPU for gloo): the packed
    moments with the local row count in the LAST slot, so the path has exactly ONE exchange step, no staging
    copy and no extra allocation.  Reading the reduced count back is the one host synchronisation the solve needs
    anyway (``n`` is a host argument of ``ccz_*_solve``); it also orders the collective before libccz's stream.
    """
    import torch
    import torch.distributed as dist

    if buf.dtype != torch.float64 or buf.dim() != 1 or buf.numel() < 2:
        raise ValueError("moments buffer must be a flat float64 tensor with the row count in its last slot")
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return int(round(float(buf[-1].item())))


def rank_and_world(group=None):
    """(rank, world size) of the enclosing ``row_sharded`` block; (0, 1) outside one."""
    import torch.distributed as dist

    if not is_sharded():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def allreduce_small(values, device, group=None):
    """SUM-reduce a small float64 NumPy array over the ranks (host bookkeeping such as per-setting scores); the
    tensor lives on ``device`` (CUDA for nccl, CPU for gloo)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    t = torch.as_tensor(np.ascontiguousarray(values, dtype=np.float64), device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()
