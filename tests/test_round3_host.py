"""CPU tests, round 3: the two-part exchange layout (host double of the C ABI), the moments-form loss oracle."""

import numpy as np

from hostsim_util import hostsim_handle


def _blocks_reference(G, s, dims):
    """NumPy statement of the blocks layout of include/ccz.h: head = [upper triangles of the diagonal blocks | s | slot],
    tail = the off-diagonal blocks (i < j), row-major."""
    off = np.concatenate([[0], np.cumsum(dims)])
    head, tail = [], []
    for i, d in enumerate(dims):
        blk = G[off[i]:off[i + 1], off[i]:off[i + 1]]
        head.append(blk[np.triu_indices(d)])
    head.append(s)
    head.append(np.zeros(1))
    for i in range(len(dims)):
        for j in range(i + 1, len(dims)):
            tail.append(G[off[i]:off[i + 1], off[j]:off[j + 1]].ravel())
    return np.concatenate(head), (np.concatenate(tail) if tail else np.zeros(0))


def test_blocks_layout_round_trip_on_the_host_double():
    H = hostsim_handle()
    rng = np.random.default_rng(0)
    for dims in ([5, 3], [4, 1, 6], [7]):
        D = sum(dims)
        X = rng.standard_normal((20, D))
        G, s = np.triu(X.T @ X), X.sum(0)                   # K1 fills the upper triangle only
        mom = np.concatenate([G.ravel(), s])
        head_ref, tail_ref = _blocks_reference(G, s, dims)
        n_head, n_tail = head_ref.size, tail_ref.size
        assert n_head + n_tail == D * (D + 1) // 2 + D + 1
        packed = np.full(n_head + n_tail, np.nan)
        H.moments_pack_blocks(mom, D, dims, packed, H.HEAD)
        assert np.array_equal(packed[:n_head - 1], head_ref[:-1]) and np.isnan(packed[n_head - 1]) and np.all(np.isnan(packed[n_head:]))
        H.moments_pack_blocks(mom, D, dims, packed, H.TAIL)
        assert np.array_equal(packed[n_head:], tail_ref)
        # two ranks' worth: the exchange is a plain sum of the packed buffers
        packed[n_head - 1] = 20.0
        summed = 2.0 * packed
        out = np.zeros_like(mom)
        H.moments_unpack_blocks(summed, D, dims, out, H.HEAD)
        o = np.concatenate([[0], np.cumsum(dims)])
        Gout = out[:D * D].reshape(D, D)
        for i in range(len(dims)):
            b = slice(o[i], o[i + 1])
            assert np.array_equal(np.triu(Gout[b, b]), 2.0 * np.triu(G[b, b]))
            for j in range(i + 1, len(dims)):
                assert np.all(Gout[b, o[j]:o[j + 1]] == 0.0)        # the tail has not arrived yet
        assert np.array_equal(out[D * D:], 2.0 * s) and summed[n_head - 1] == 40.0
        H.moments_unpack_blocks(summed, D, dims, out, H.TAIL)
        assert np.array_equal(np.triu(out[:D * D].reshape(D, D)), 2.0 * G)
        both = np.zeros_like(packed)
        H.moments_pack_blocks(mom, D, dims, both, H.BOTH)
        assert np.array_equal(both[:n_head - 1], head_ref[:-1]) and np.array_equal(both[n_head:], tail_ref)


def test_loss_from_moments_equals_the_closed_form():
    from oracle import losses as ol

    rng = np.random.default_rng(1)
    z1 = rng.standard_normal((400, 11)) + 3.0
    z2 = 0.6 * z1[:, :7] + rng.standard_normal((400, 7))
    l, g1, g2 = ol.cca_loss_closed_form(z1, z2, 1e-4)
    Z = np.hstack([z1, z2])
    l2, Gamma, mean = ol.cca_loss_from_moments(Z.T @ Z, Z.sum(0), 400, 11, 7, 1e-4)
    assert abs(l - l2) < 1e-12 * abs(l)
    g = (Z - mean) @ Gamma
    assert np.abs(g[:, :11] - g1).max() < 1e-12 and np.abs(g[:, 11:] - g2).max() < 1e-12


def test_bench_solution_gate_accepts_fits_and_rejects_perturbed_ones(monkeypatch):
    """bench.py's gate of the extras (pencil certificate on the moments of the timed views) on the host double: the
    estimators' own solutions pass for rCCA / MCCA / GCCA, the same models with one weight column disturbed -- or with
    the two leading columns swapped -- fail.  (On the GPU the gate runs on device moments; the logic is the same code.)"""
    import bench
    from cca_zoo_amd import _backend
    from cca_zoo_amd.linear import GCCA, MCCA, rCCA
    from oracle import reference_form as rf

    h = hostsim_handle()
    monkeypatch.setattr(_backend, "default_handle", lambda device=None: h)
    monkeypatch.setattr(_backend, "handle_for", lambda arrays: h)
    views = rf.joint_data(3, 400, 4, [14, 11, 9], 2.0, 3)
    cases = (("rcca", rCCA(latent_dimensions=3, c=0.1), views[:2], 0.1),
             ("mcca", MCCA(latent_dimensions=3, c=0.2), views, 0.2),
             ("gcca", GCCA(latent_dimensions=3, c=0.1), views, 0.1))
    for est, model, vs, c in cases:
        model.fit(vs)
        good = bench.solution_gate(model, vs, est, c)
        assert good["ok"] and good["pencil_eigenvalues_above_lambda_k"] == 3, (est, good)
        keep = [w.copy() for w in model.weights_]
        model.weights_[0][:, 1] *= 1.0 + 1e-3                      # a slightly wrong direction / normalisation
        assert not bench.solution_gate(model, vs, est, c)["ok"], est
        model.weights_ = [w[:, [1, 0, 2]].copy() for w in keep]    # right subspace, wrong order of the pairs
        assert not bench.solution_gate(model, vs, est, c)["ok"], est
        model.weights_ = keep
