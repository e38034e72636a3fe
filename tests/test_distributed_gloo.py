"""world_size-2 gloo test of the sharded path's host logic (no GPU).

Each rank owns a row shard, forms its partial moments (here with NumPy as the checker
-- on the GPU the same buffer comes out of K1), runs the ONE collective of the path
(``allreduce_moments``) and then the product solver drivers (host test double) on the
reduced moments.  Every rank must reproduce the single-process result.
"""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cca_zoo_amd import _dist
        from hostsim_util import hostsim_handle
        from oracle import gram_form as gf
        from oracle import reference_form as rf

        views = rf.joint_data(2, 301, 3, [12, 9], 2.0, 5)       # same data on every rank
        lo, hi = _dist.shard_bounds(301, rank, world)
        G, s, n_loc = gf.moments([v[lo:hi] for v in views])
        buf = torch.from_numpy(np.concatenate([G.ravel(), s, [float(n_loc)]]))     # row count in the tail slot
        with _dist.row_sharded():
            assert _dist.is_sharded()
            n_tot = _dist.allreduce_moments(buf, _dist.active_group())
        mom = buf[:-1]
        assert not _dist.is_sharded()
        H = hostsim_handle()
        W, means, vals = H.rcca_solve(mom.numpy(), n_tot, [12, 9], [0.1, 0.1], True, 3)
        q.put((rank, n_tot, [w.copy() for w in W], [m.copy() for m in means]))
    finally:
        dist.destroy_process_group()


def test_shard_bounds_partition():
    from cca_zoo_amd import shard_bounds

    for n in (0, 1, 7, 100, 1_000_003):
        for w in (1, 2, 3, 8):
            edges = [shard_bounds(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_row_sharded_requires_process_group():
    from cca_zoo_amd import row_sharded

    with pytest.raises(RuntimeError, match="process group"):
        with row_sharded():
            pass


def test_two_rank_allreduce_then_solve():
    from conftest import col_rel_err
    from oracle import reference_form as rf

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    views = rf.joint_data(2, 301, 3, [12, 9], 2.0, 5)
    W_ref, means_ref = rf.rcca_weights(views, 3, c=0.1)
    for rank, n_tot, W, means in results:
        assert n_tot == 301
        for w, r in zip(W, W_ref):
            assert col_rel_err(w, r) < 1e-9
        for m, r in zip(means, means_ref):
            np.testing.assert_allclose(m, r, atol=1e-12)


def _estimator_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cca_zoo_amd import _backend, row_sharded, shard_bounds
        from cca_zoo_amd.linear import GCCA, MCCA, rCCA
        from hostsim_util import hostsim_handle
        from oracle import reference_form as rf

        h = hostsim_handle()
        _backend.default_handle = lambda device=None: h
        views = rf.joint_data(3, 257, 3, [10, 8, 6], 2.0, 9)       # same data on every rank
        lo, hi = shard_bounds(257, rank, world)
        local = [v[lo:hi] for v in views]
        out = {}
        with row_sharded():
            m = rCCA(latent_dimensions=2, c=0.1).fit(local[:2])
            out["rcca"] = ([w.copy() for w in m.weights_], [x.copy() for x in m.means_], m.n_samples_,
                           m.score(local[:2]).copy())
            mm = MCCA(latent_dimensions=2, c=0.2).fit(local)
            out["mcca"] = [w.copy() for w in mm.weights_]
            gg = GCCA(latent_dimensions=2, c=0.2).fit(local)
            out["gcca"] = [w.copy() for w in gg.weights_]
            from cca_zoo_amd.linear import GRCCA, PartialCCA

            conf = np.random.default_rng(3).standard_normal((257, 2)) + 0.4        # same confounds on every rank
            pc = PartialCCA(latent_dimensions=2, c=0.1).fit(local[:2], partials=conf[lo:hi])
            out["pcca"] = ([w.copy() for w in pc.weights_], [b.copy() for b in pc.confound_betas_])
            groups = [np.arange(10) // 3, np.arange(8) // 4]
            gr = GRCCA(latent_dimensions=2, c=[0.5, 0.3], mu=[0.2, 0.0]).fit(local[:2], feature_groups=groups)
            out["grcca"] = [w.copy() for w in gr.weights_]
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_estimators_match_single_process():
    """The estimators themselves inside row_sharded() (K1 on the local rows through the host double, packed
    all-reduce over gloo, replicated solve): every rank reproduces the fit on the full data, and score() inside
    the context is the global-sample score."""
    from conftest import col_rel_err
    from oracle import reference_form as rf

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_estimator_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    views = rf.joint_data(3, 257, 3, [10, 8, 6], 2.0, 9)
    Wr, mr = rf.rcca_weights(views[:2], 2, c=0.1)
    Wm, _ = rf.mcca_weights(views, 2, c=0.2)
    Wg, _ = rf.gcca_weights(views, 2, c=0.2)
    score_ref = rf.mean_offdiag_corr(views[:2], Wr, mr)
    from oracle import partial_group as pg

    conf = np.random.default_rng(3).standard_normal((257, 2)) + 0.4
    Wp, _, Bp = pg.partialcca_reference_form(views[:2], conf, 2, c=0.1)
    Wgr, _ = pg.grcca_reference_form(views[:2], [np.arange(10) // 3, np.arange(8) // 4], 2, c=[0.5, 0.3], mu=[0.2, 0.0])
    for rank, out in results:
        W, means, n_seen, score = out["rcca"]
        assert n_seen == 257
        for w, r in zip(W, Wr):
            assert col_rel_err(w, r) < 1e-8
        for a, b in zip(means, mr):
            np.testing.assert_allclose(a, b, atol=1e-12)
        np.testing.assert_allclose(score, score_ref, rtol=1e-8, atol=1e-10)
        for w, r in zip(out["mcca"], Wm):
            assert col_rel_err(w, r) < 1e-7
        for w, r in zip(out["gcca"], Wg):
            assert col_rel_err(w, r) < 1e-7
        for w, r in zip(out["pcca"][0], Wp):
            assert col_rel_err(w, r) < 1e-7
        for b, r in zip(out["pcca"][1], Bp):
            np.testing.assert_allclose(b, r, rtol=1e-8, atol=1e-10)
        for w, r in zip(out["grcca"], Wgr):
            assert col_rel_err(w, r) < 1e-7


def _grid_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cca_zoo_amd import _backend, row_sharded, shard_bounds
        from cca_zoo_amd.linear import rCCA
        from cca_zoo_amd.model_selection import GridSearchCV
        from hostsim_util import hostsim_handle
        from oracle import reference_form as rf

        h = hostsim_handle()
        _backend.default_handle = lambda device=None: h
        views = rf.joint_data(2, 203, 3, [9, 7], 1.5, 4)
        lo, hi = shard_bounds(203, rank, world)
        with row_sharded():
            gs = GridSearchCV(rCCA(latent_dimensions=2), {"c": [0.01, 0.2, 0.7]}, cv=3).fit([v[lo:hi] for v in views])
        q.put((rank, gs.route_, np.stack([gs.cv_results_[f"split{f}_test_score"] for f in range(3)], axis=1),
               gs.best_params_, [w.copy() for w in gs.best_estimator_.weights_], gs.best_estimator_.n_samples_))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_grid_search():
    """GridSearchCV inside row_sharded(): folds are cut within every shard, each fold's moments are all-reduced, every
    rank runs the same solves.  Equal to a single-process search whose splitter assigns the same global folds."""
    from sklearn.model_selection import KFold, PredefinedSplit

    from conftest import col_rel_err
    from hostsim_util import hostsim_handle
    from oracle import reference_form as rf

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    from cca_zoo_amd import _backend, shard_bounds
    from cca_zoo_amd.linear import rCCA
    from cca_zoo_amd.model_selection import GridSearchCV

    views = rf.joint_data(2, 203, 3, [9, 7], 1.5, 4)
    fold_of_row = np.empty(203, dtype=int)
    for r in range(2):
        lo, hi = shard_bounds(203, r, 2)
        for f, (_, te) in enumerate(KFold(3).split(np.zeros((hi - lo, 1)))):
            fold_of_row[lo + te] = f
    h = hostsim_handle()
    saved = _backend.default_handle
    _backend.default_handle = lambda device=None: h
    try:
        ref = GridSearchCV(rCCA(latent_dimensions=2), {"c": [0.01, 0.2, 0.7]}, cv=PredefinedSplit(fold_of_row)).fit(views)
    finally:
        _backend.default_handle = saved
    ref_scores = np.stack([ref.cv_results_[f"split{f}_test_score"] for f in range(3)], axis=1)
    for rank, route, scores, best, W, n_seen in results:
        assert route == "shared-moments" and n_seen == 203
        np.testing.assert_allclose(scores, ref_scores, rtol=1e-9, atol=1e-11)
        assert best == ref.best_params_
        for w, r in zip(W, ref.best_estimator_.weights_):
            assert col_rel_err(w, r) < 1e-8
