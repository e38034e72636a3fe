"""GPU tests of the N > 1 path that need TWO visible GPUs (skipped on the one-GPU boxes of this pool; they execute on the
first multi-GPU box): two processes, one per GPU, fit rCCA / MCCA / GCCA on their row shards through
 * libccz's own RCCL communicator (``CczComm.from_file`` -> ``ccz_moments_exchange``), and
 * ``torch.distributed`` (backend nccl = RCCL) with the two-part overlapped exchange,
and rank 0 compares every result with the single-process fit of the whole data at 1e-10.  Plus, on any GPU box: the
whole-exchange entry at world size one and the "librccl cannot be found" path.

Reference seams: the fits of cca_zoo/linear/_rcca.py:69-101, _mcca.py:99-197, _gcca.py:80-110 on row shards."""

import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    try:
        import torch

        return torch.cuda.device_count()
    except Exception:
        return 0


_RANK = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch
rank, world, transport = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
torch.cuda.set_device(rank)
from cca_zoo_amd import _backend, _dist, row_sharded, shard_bounds
from cca_zoo_amd.linear import GCCA, MCCA, rCCA
from oracle import reference_form as rf
h = _backend.default_handle(rank)
if transport == "ccz":
    group = _dist.CczComm.from_file(sys.argv[4], world, rank, handle=h, tag=sys.argv[5])
else:
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[5]
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    group = None
views = rf.joint_data(3, 6001, 4, [300, 260, 200], 2.0, 21)          # the same data in every rank; 6001 rows: ragged shards
lo, hi = shard_bounds(6001, rank, world)
ok = True
for name, make, vs in (("rcca", lambda: rCCA(latent_dimensions=4, c=[0.1, 0.2]), views[:2]),
                       ("mcca", lambda: MCCA(latent_dimensions=4, c=0.1), views),
                       ("gcca", lambda: GCCA(latent_dimensions=4, c=0.05, view_weights=[1.0, 2.0, 0.5]), views)):
    for on_device in (False, True):
        mine = [v[lo:hi] for v in vs]
        if on_device:
            mine = [torch.as_tensor(v, device=f"cuda:{{rank}}") for v in mine]
        with row_sharded(group=group):
            m = make().fit(mine)
            sc = m.score(mine)
        if rank == 0:
            ref = make().fit(vs)
            for a, b in zip(ref.weights_, m.weights_):
                err = float(np.abs(np.asarray(b) - a).max() / np.abs(a).max())
                ok = ok and err < 1e-10
                print(name, "device" if on_device else "host", "weights", err, flush=True)
            e2 = float(np.abs(np.asarray(sc) - ref.score(vs)).max())
            ok = ok and e2 < 1e-10
            print(name, "score", e2, flush=True)
if transport == "ccz":
    assert h.comm_info() == (world, rank)
    group.close()
else:
    dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok" if ok else "FAILED", flush=True)
sys.exit(0 if ok else 1)
"""


def _run_ranks(transport):
    world = 2
    with tempfile.TemporaryDirectory() as td:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        script = os.path.join(td, "rank.py")
        with open(script, "w") as f:
            f.write(_RANK.format(root=ROOT))
        procs = [subprocess.Popen([sys.executable, script, str(r), str(world), transport, os.path.join(td, "comm.id"), str(port)],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = [p.communicate(timeout=900)[0] for p in procs]
        for r, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and f"rank {r} ok" in o, f"rank {r}:\n{o[-4000:]}"


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs")
def test_two_ranks_over_the_ccz_communicator_match_the_single_process_fit():
    _run_ranks("ccz")


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs")
def test_two_ranks_over_torch_distributed_nccl_match_the_single_process_fit():
    _run_ranks("torch")


def test_whole_exchange_entry_world_size_one_and_buffer_reuse():
    """ccz_moments_exchange with ONE rank is the identity on the moments (pack -> all-reduce -> unpack), returns the row
    count, leaves the tail registered for the next solve, and reuses the handle's buffer across sizes and fits."""
    import torch

    from cca_zoo_amd import _backend, _dist, row_sharded
    from cca_zoo_amd.linear import MCCA, rCCA
    from oracle import reference_form as rf

    H = _backend.default_handle(0)
    comm = _dist.CczComm(H, H.comm_unique_id(), 1, 0)
    try:
        rng = np.random.default_rng(3)
        for dims in ([40, 24], [16, 8, 12], [40, 24]):
            D = sum(dims)
            G = rng.standard_normal((D, D)); G = G @ G.T
            mom = np.concatenate([np.triu(G).reshape(-1), rng.standard_normal(D)])
            buf = H.to_device(mom)
            assert H.moments_exchange(buf.ptr, D, dims, 1234) == 1234
            H.solve_defer(None)                               # not a solve: await the tail on the handle's stream (ccz.h) ...
            H.sync()                                          # ... and that stream alone is what the host waits for
            got = H.to_host(buf, (D * D + D,))
            np.testing.assert_array_equal(np.triu(got[:D * D].reshape(D, D)), np.triu(G))
            np.testing.assert_array_equal(got[D * D:], mom[D * D:])
        views = rf.joint_data(3, 3000, 4, [96, 80, 72], 2.0, 9)
        for make, vs in ((lambda: rCCA(latent_dimensions=4, c=0.1), views[:2]), (lambda: MCCA(latent_dimensions=4, c=0.1), views)):
            plain = make().fit(vs)
            for _ in range(3):                              # the same buffer three times, the deferred tail consumed by every solve
                with row_sharded(group=comm):
                    m = make().fit([torch.as_tensor(v, device="cuda") for v in vs])
                for a, b in zip(plain.weights_, m.weights_):
                    np.testing.assert_allclose(b, a, rtol=1e-10, atol=1e-12)
    finally:
        comm.close()


def test_missing_rccl_is_reported_not_crashed():
    """CCZ_RCCL_LIB points at nothing: the first communicator call returns CCZ_ERCCL with dlopen's message (the
    message used to be built from a second, NULL, dlerror())."""
    code = f"""
import os, sys
os.environ['CCZ_RCCL_LIB'] = '/nonexistent/librccl.so.1'
sys.path.insert(0, {ROOT!r})
from cca_zoo_amd import _backend
h = _backend.default_handle(0)
try:
    h.comm_unique_id()
except RuntimeError as e:
    assert 'librccl not found' in str(e) and 'nonexistent' in str(e), str(e)
    print('reported')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "reported" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
