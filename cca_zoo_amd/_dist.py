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


class CczComm:
    """A row-sharding group that lives entirely behind libccz's C ABI (``ccz_comm_*`` / ``ccz_allreduce_sum_f64``: RCCL
    over xGMI, one process per GPU) -- for callers of the linear path that do not run ``torch.distributed``.

    ``CczComm(handle, unique_id, world, rank)`` joins the communicator (rank 0 obtains ``unique_id`` from
    ``handle.comm_unique_id()`` and ships the 128 bytes to the other ranks); ``CczComm.from_file(path, world, rank)`` does
    the shipping through a file on a shared file system.  Pass it to ``row_sharded(group=comm)``.
    """

    def __init__(self, handle, unique_id, world, rank):
        self.handle, self.world, self.rank = handle, int(world), int(rank)
        handle.comm_init_rank(unique_id, world, rank)

    @classmethod
    def from_file(cls, path, world, rank, handle=None, timeout_s=120.0, tag=None):
        """Ship the communicator id through ``path`` on a file system every rank sees.  The file carries a 32-byte run tag
        behind the 128-byte id: ``tag`` (any string; default: the launcher's ``MASTER_ADDR:MASTER_PORT:TORCHELASTIC_RUN_ID``
        when a run id is present -- without one a ``tag`` is required) -- readers ignore a file whose tag is not theirs, so an id left behind by an earlier run is never
        joined.  Rank 0 removes a stale file first and deletes its own in :meth:`close`.  Without a launcher and without
        ``tag`` the path must be fresh per run."""
        import hashlib
        import os
        import time

        from cca_zoo_amd import _backend

        h = handle or _backend.default_handle()
        if tag is None:
            run_id = os.environ.get("TORCHELASTIC_RUN_ID", "")
            if run_id in ("", "none"):
                # a static rendezvous (fixed port, no run id) gives the SAME default tag to every run: a rank could join the
                # id file a crashed run left behind and hang in ncclCommInitRank -- ask for a tag instead
                raise ValueError("CczComm.from_file: no per-run launcher id in the environment (TORCHELASTIC_RUN_ID); pass tag=<a string "
                                 "unique to this run> or use a fresh path")
            tag = ":".join((os.environ.get("MASTER_ADDR", ""), os.environ.get("MASTER_PORT", ""), run_id))
        mark = hashlib.sha256(str(tag).encode()).digest()
        if rank == 0:
            try:
                os.unlink(path)                            # an id of an earlier run must not be read before ours lands
            except FileNotFoundError:
                pass
            uid = h.comm_unique_id()
            tmp = f"{path}.tmp.{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(uid + mark)
            os.replace(tmp, path)                          # atomic: readers never see a partial id
        else:
            t0 = time.time()
            uid = None
            while uid is None:
                try:
                    with open(path, "rb") as f:
                        blob = f.read()
                    if len(blob) == 160 and blob[128:] == mark:
                        uid = blob[:128]
                except FileNotFoundError:
                    pass
                if uid is None:
                    if time.time() - t0 > timeout_s:
                        raise TimeoutError(f"no communicator id for this run at {path} after {timeout_s:.0f} s")
                    time.sleep(0.02)
        comm = cls(h, uid, world, rank)
        comm._id_file = path if rank == 0 else None
        return comm

    def close(self):
        self.handle.comm_destroy()
        path = getattr(self, "_id_file", None)
        if path:
            import os

            try:
                os.unlink(path)
            except OSError:
                pass
            self._id_file = None

    def allreduce_small(self, values):
        import numpy as np

        a = np.ascontiguousarray(values, dtype=np.float64)
        buf = self.handle.to_device(a.reshape(-1))
        self.handle.allreduce_sum_f64(buf.ptr, a.size)
        return self.handle.to_host(buf, a.shape)


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
    if not isinstance(group, CczComm):
        import torch.distributed as dist

        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("row_sharded() needs an initialised torch.distributed process group or a CczComm")
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


def staged_over_gloo(t, group=None) -> bool:
    """True for a CUDA tensor reduced over a ``gloo`` group: the collective then runs on a pinned host copy (d2h ->
    gloo all-reduce -> h2d).  A TEST transport -- RCCL refuses two ranks on one device, gloo does not care -- that lets two
    processes share ONE GPU and still run every rank-dependent line of the sharded path with the real kernels
    (tests/test_gpu_two_ranks_one_gpu.py, ``bench.py --transport gloo-staged``).  Production groups are nccl (= RCCL)."""
    import torch.distributed as dist

    return bool(getattr(t, "is_cuda", False)) and dist.get_backend(group) == "gloo"


def all_reduce_sum(t, group=None):
    """SUM all-reduce of a torch tensor in place over a torch.distributed group (CUDA over gloo: staged, see above)."""
    import torch
    import torch.distributed as dist

    if staged_over_gloo(t, group):
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        host.copy_(t)                                   # waits for the producing stream
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()       # `host` may go once the copy has run
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def allreduce_moments(buf, group=None):
    """SUM-reduce ``buf`` IN PLACE and return the global row count.

    ``buf`` is a flat float64 torch tensor ``[payload ... | n_local]`` (CUDA for nccl, CPU for gloo): the packed
    moments with the local row count in the LAST slot, so the path has exactly ONE exchange step, no staging
    copy and no extra allocation.  Reading the reduced count back is the one host synchronisation the solve needs
    anyway (``n`` is a host argument of ``ccz_*_solve``); it also orders the collective before libccz's stream.
    """
    import torch

    if buf.dtype != torch.float64 or buf.dim() != 1 or buf.numel() < 2:
        raise ValueError("moments buffer must be a flat float64 tensor with the row count in its last slot")
    if isinstance(group, CczComm):
        # the collective behind the C ABI (the sharded losses inside row_sharded(group=CczComm)): libccz's stream follows
        # torch's current stream on the device, the all-reduce runs on it, torch's stream follows back
        if not buf.is_cuda:
            raise ValueError("a CczComm reduces device buffers")
        sp = int(torch.cuda.current_stream(buf.device).cuda_stream)
        h = group.handle
        h.acquire(sp)
        h.allreduce_sum_f64(buf.data_ptr(), buf.numel())
        h.release(sp)
        return int(round(float(buf[-1].item())))
    all_reduce_sum(buf, group)
    return int(round(float(buf[-1].item())))


def rank_and_world(group=None):
    """(rank, world size) of the enclosing ``row_sharded`` block; (0, 1) outside one."""
    if not is_sharded():
        return 0, 1
    if isinstance(group, CczComm):
        return group.rank, group.world
    import torch.distributed as dist

    return dist.get_rank(group), dist.get_world_size(group)


def allreduce_small(values, device, group=None):
    """SUM-reduce a small float64 NumPy array over the ranks (host bookkeeping such as per-setting scores); the
    tensor lives on ``device`` (CUDA for nccl, CPU for gloo)."""
    if isinstance(group, CczComm):
        return group.allreduce_small(values)
    import numpy as np
    import torch
    import torch.distributed as dist

    t = torch.as_tensor(np.ascontiguousarray(values, dtype=np.float64), device=device)
    all_reduce_sum(t, group)
    return t.cpu().numpy()
