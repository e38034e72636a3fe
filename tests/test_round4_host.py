"""CPU tests of round 4's host logic: bench.py's self-launch, the C-ABI collective's Python route (CczComm) on the host
double, the torch-less load of libccz, the exported comm symbols."""

import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from hostsim_util import hostsim_handle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1, free port) and still prints ONE JSON line -- here with gloo and no GPU work."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-test"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out == {"launch_test": True, "n_gpus": 2, "ranks": 2, "sum": 3.0}


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-test"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_comm_symbols_are_exported_and_bound():
    from cca_zoo_amd import _backend

    lib = _backend.bind(C.CDLL(_backend.library_path()), strict=True)
    for name in ("ccz_comm_unique_id", "ccz_comm_init_rank", "ccz_comm_init_all", "ccz_comm_info", "ccz_comm_destroy",
                 "ccz_allreduce_sum_f64", "ccz_allreduce_sum_f64_multi"):
        assert hasattr(lib, name)
        assert name in _backend.SIGNATURES
    header = open(os.path.join(ROOT, "include", "ccz.h")).read()
    assert "ccz_allreduce_sum_f64(ccz_handle h, double* buf_dev, int64_t count)" in header


def test_ccz_comm_route_of_row_sharded_on_the_host_double(monkeypatch):
    """row_sharded(group=CczComm): pack -> ccz_allreduce_sum_f64 -> unpack -> solve, without torch.distributed.  World
    size one on the host double (its collective is a no-op): the fit must equal the unsharded fit bit for bit, the
    timings record the exchange, and a bigger world is refused by the double (it has no transport)."""
    from cca_zoo_amd import _backend, _dist, row_sharded
    from cca_zoo_amd.linear import MCCA, rCCA
    from oracle import reference_form as rf

    h = hostsim_handle()
    monkeypatch.setattr(_backend, "default_handle", lambda device=None: h)
    views = rf.joint_data(3, 240, 3, [14, 11, 9], 2.0, 3)
    uid = h.comm_unique_id()
    assert len(uid) == 128
    with pytest.raises(RuntimeError, match="transport"):
        _dist.CczComm(h, uid, 2, 0)
    comm = _dist.CczComm(h, uid, 1, 0)
    assert h.comm_info() == (1, 0)
    consumed = []
    real_defer = h.solve_defer
    monkeypatch.setattr(h, "solve_defer", lambda ev: (consumed.append(ev), real_defer(ev))[1])
    try:
        for make in (lambda: rCCA(latent_dimensions=3, c=0.2), lambda: MCCA(latent_dimensions=3, c=0.1)):
            vs = views[:2] if isinstance(make(), rCCA) else views
            plain = make().fit(vs)
            with row_sharded(group=comm):
                assert _dist.is_sharded() and _dist.rank_and_world(comm) == (0, 1)
                sharded = make().fit(vs)
                assert consumed == []                       # a fit leaves the exchange's tail to its solve (overlap) ...
                sc = sharded.score(vs)
                assert consumed and all(ev is None for ev in consumed)   # ... every other reader of the moments awaits it first
                consumed.clear()
            for a, b in zip(plain.weights_, sharded.weights_):
                np.testing.assert_array_equal(a, b)
            np.testing.assert_allclose(sc, plain.score(vs), rtol=1e-12)
            assert sharded.n_samples_ == 240
    finally:
        comm.close()
    assert h.comm_info() == (0, -1)
    with pytest.raises(ValueError, match="no communicator"):
        h.allreduce_sum_f64(h.alloc(64).ptr, 8)


def test_torchless_load_does_not_import_torch():
    """CCZ_TORCHLESS=1: a ctypes-only caller of the linear path loads libccz on /opt/rocm's runtime without torch
    (VERDICT r3 item 10).  The default still maps torch's runtime first (one HIP runtime per process)."""
    code = ("import os, sys; os.environ['CCZ_TORCHLESS'] = '1'; sys.path.insert(0, %r); "
            "from cca_zoo_amd import _backend; lib = _backend.library(); "
            "assert lib.ccz_version() >= 130; assert 'torch' not in sys.modules; print('ok')" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stderr[-2000:]


def test_bench_weights_agreement_separated_and_clustered_columns():
    """bench.py's comparator of two solutions: per column where the neighbouring correlations are
    separated by more than 100 x tol, as a subspace where they are not -- a rotation INSIDE a cluster must pass, a
    perturbed separated column and a wrong subspace must fail."""
    import bench

    rng = np.random.default_rng(0)
    d, k = 40, 6
    W = np.linalg.qr(rng.standard_normal((d, k)))[0]
    vals = np.array([0.9, 0.7, 0.5, 0.3000, 0.3000 + 1e-7, 0.1])             # columns 3 and 4 form a cluster
    th = 0.7
    Rm = np.eye(k)
    Rm[3:5, 3:5] = [[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]
    ok = bench.weights_agreement([W @ Rm * np.array([1, -1, 1, 1, 1, -1.0])], [W], vals, 1e-3)
    assert ok["ok"] and ok["separated_columns"] == 4 and ok["max_col_rel_err_separated"] < 1e-12 and ok["subspace_residual"] < 1e-12
    W2 = W.copy()
    W2[:, 1] += 0.01 * rng.standard_normal(d)                                  # a separated column is off
    bad = bench.weights_agreement([W2], [W], vals, 1e-3)
    assert not bad["ok"] and bad["max_col_rel_err_separated"] > 1e-3
    W3 = W.copy()
    W3[:, 3] = rng.standard_normal(d)                                          # the cluster leaves the subspace
    assert not bench.weights_agreement([W3], [W], vals, 1e-3)["ok"]


def test_ccz_comm_from_file_ignores_a_stale_id_and_cleans_up(tmp_path, monkeypatch):
    """an id file left behind by an earlier run must never be joined.  The file carries a run tag; a reader polls
    until an id with ITS tag appears, rank 0 removes whatever was there before it writes, and ``close()`` deletes the file.
    (The host double has no transport beyond world size one, so the communicator's constructor is stubbed; the file protocol
    is the code under test.)"""
    import threading
    import time

    from cca_zoo_amd import _dist

    h = hostsim_handle()
    made = []

    def fake_init(self, handle, unique_id, world, rank):
        self.handle, self.world, self.rank = handle, world, rank
        made.append((rank, bytes(unique_id)))

    monkeypatch.setattr(_dist.CczComm, "__init__", fake_init)
    monkeypatch.setattr(h, "comm_destroy", lambda: None)
    path = str(tmp_path / "comm.id")
    with open(path, "wb") as f:                       # a previous run's id: right length, another tag
        f.write(b"\x11" * 128 + b"\x22" * 32)
    got = {}

    def reader():
        got["comm"] = _dist.CczComm.from_file(path, 2, 1, handle=h, timeout_s=20.0, tag="run-2")

    t = threading.Thread(target=reader)
    t.start()
    time.sleep(0.3)
    assert t.is_alive() and not made                  # the stale file is there and is NOT accepted
    monkeypatch.setattr(h, "comm_unique_id", lambda: b"\x5a" * 128)
    c0 = _dist.CczComm.from_file(path, 2, 0, handle=h, tag="run-2")
    t.join(timeout=20.0)
    assert not t.is_alive()
    assert sorted(made) == [(0, b"\x5a" * 128), (1, b"\x5a" * 128)]
    assert os.path.exists(path)
    got["comm"].close()                               # a reader does not own the file
    assert os.path.exists(path)
    c0.close()
    assert not os.path.exists(path)
    with pytest.raises(TimeoutError, match="for this run"):
        _dist.CczComm.from_file(path, 2, 1, handle=h, timeout_s=0.2, tag="run-3")
