"""CPU: randomized stress of the product solver drivers (csrc/solve.cpp on the host double) against the
second-moment oracle -- odd view widths, k up to the smallest width, ridge 0 / small / 1, centring on and off,
more views, both the direct (small p) and the Chebyshev (p > 192) eigen-solver routes."""

import numpy as np
import pytest

from conftest import col_rel_err
from hostsim_util import hostsim_handle, pack_moments
from oracle import gram_form as gf


@pytest.fixture(scope="module")
def H():
    return hostsim_handle()


def _data(rng, n, dims, latent=3, noise=0.7):
    z = rng.standard_normal((n, latent)) * np.linspace(2.0, 0.8, latent)
    return [z @ rng.standard_normal((latent, d)) + noise * rng.standard_normal((n, d)) + rng.standard_normal(d)
            for d in dims]


def _subspace_gap_ok(vals, k):
    """Per-column comparison needs separated leading values."""
    v = np.asarray(vals)
    return len(v) < 2 or np.min(np.abs(np.diff(v[:k]))) > 1e-6 * max(1.0, abs(v[0]))


@pytest.mark.parametrize("seed", range(12))
def test_rcca_random_shapes(H, seed):
    rng = np.random.default_rng(100 + seed)
    d1, d2 = int(rng.integers(1, 24)), int(rng.integers(1, 24))
    n = int(rng.integers(max(d1, d2) + 5, 200))
    k = int(rng.integers(1, min(d1, d2) + 1))
    c = [float(rng.choice([0.0, 1e-3, 0.2, 1.0])), float(rng.choice([0.0, 0.05, 1.0]))]
    center = bool(rng.integers(0, 2))
    views = _data(rng, n, [d1, d2])
    G, s, _ = gf.moments(views)
    W, means, vals = H.rcca_solve(pack_moments(G, s), n, [d1, d2], c, center, k)
    Wr, mr, vr = gf.rcca_from_moments(G, s, n, [d1, d2], k, c=tuple(c), center=center)
    np.testing.assert_allclose(vals, vr[: len(vals)], rtol=1e-8, atol=1e-10)
    if _subspace_gap_ok(vr, k):
        for w, r in zip(W, Wr):
            assert col_rel_err(w, r) < 1e-6, (d1, d2, n, k, c, center)
    for a, b in zip(means, mr):
        np.testing.assert_allclose(a, b, atol=1e-12)


@pytest.mark.parametrize("seed", range(10))
def test_mcca_gcca_random_shapes(H, seed):
    rng = np.random.default_rng(300 + seed)
    m = int(rng.integers(2, 5))
    dims = [int(rng.integers(2, 14)) for _ in range(m)]
    n = int(rng.integers(sum(dims) + 10, 260))
    k = int(rng.integers(1, min(dims) + 1))
    c = [float(rng.choice([1e-3, 0.1, 0.5])) for _ in range(m)]
    center = bool(rng.integers(0, 2))
    views = _data(rng, n, dims, latent=4)
    G, s, _ = gf.moments(views)
    mom = pack_moments(G, s)
    W, _, vals = H.mcca_solve(mom, n, dims, c, 1e-6, center, k)
    Wr, _, vr = gf.mcca_from_moments(G, s, n, dims, k, c=c, eps=1e-6, center=center)
    np.testing.assert_allclose(vals, vr[: len(vals)], rtol=1e-8, atol=1e-10)
    if _subspace_gap_ok(vr, k):
        for w, r in zip(W, Wr):
            assert col_rel_err(w, r) < 1e-6, ("mcca", dims, n, k, c, center)
    mu = [float(rng.choice([0.5, 1.0, 2.0])) for _ in range(m)]
    W, _, vals = H.gcca_solve(mom, n, dims, c, mu, 1e-6, center, k)
    Wr, _, vr = gf.gcca_from_moments(G, s, n, dims, k, c=c, view_weights=mu, eps=1e-6, center=center)
    np.testing.assert_allclose(vals, vr[: len(vals)], rtol=1e-7, atol=1e-9)
    if _subspace_gap_ok(vr, k):
        for w, r in zip(W, Wr):
            assert col_rel_err(w, r) < 1e-6, ("gcca", dims, n, k, c, mu, center)


@pytest.mark.parametrize("d1,d2,k", [(230, 210, 5), (260, 40, 12)])
def test_rcca_chebyshev_route(H, d1, d2, k):
    """p > 192 and 3k < p: the subspace iteration (not the direct Jacobi) produces the singular triplets."""
    rng = np.random.default_rng(d1 + d2)
    n = 900
    views = _data(rng, n, [d1, d2], latent=k + 3, noise=1.0)
    G, s, _ = gf.moments(views)
    W, _, vals = H.rcca_solve(pack_moments(G, s), n, [d1, d2], [0.3, 0.3], True, k)
    Wr, _, vr = gf.rcca_from_moments(G, s, n, [d1, d2], k, c=(0.3, 0.3), center=True)
    np.testing.assert_allclose(vals, vr[:k], rtol=1e-9, atol=1e-11)
    for w, r in zip(W, Wr):
        assert col_rel_err(w, r) < 1e-6
