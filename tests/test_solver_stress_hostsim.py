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


@pytest.mark.parametrize("dims,k,c", [
    ([120, 100, 90], 6, [0.95, 0.0, 0.5]),      # ridge close to 1: the proven lower bound -1 / (1 - c) is loose but valid
    ([150, 130], 5, [0.0, 0.0]),                # two views, c = 0: spectrum in [-1, 1] with exact +/- pairs
    ([80, 70, 60, 50], 8, [1.0, 0.2, 0.2, 0.2]),  # c = 1 on one view: no closed-form bound, falls back to -||S||_inf
])
def test_mcca_gcca_chebyshev_route_with_spectrum_bounds(H, dims, k, c):
    """D > 192 and 3k < D: MCCA / GCCA go through the Chebyshev-filtered subspace iteration, whose damped interval now
    starts at a PROVEN lower bound of the spectrum (MCCA: -max 1/(1-c_i); GCCA: the Gram form is PSD) instead of
    -||S||_inf.  A bound that was not a bound would damp wanted directions: compare with the dense oracle."""
    rng = np.random.default_rng(sum(dims) + k)
    n = 1200
    views = _data(rng, n, dims, latent=k + 2, noise=1.0)
    G, s, _ = gf.moments(views)
    mom = pack_moments(G, s)
    W, _, vals = H.mcca_solve(mom, n, dims, c, 1e-6, True, k)
    Wr, _, vr = gf.mcca_from_moments(G, s, n, dims, k, c=c, eps=1e-6, center=True)
    np.testing.assert_allclose(vals, vr[:k], rtol=1e-9, atol=1e-11)
    if _subspace_gap_ok(vr, k):
        for w, r in zip(W, Wr):
            assert col_rel_err(w, r) < 1e-6, ("mcca", dims, c)
    cg = [min(ci, 0.9) for ci in c]
    W, _, vals = H.gcca_solve(mom, n, dims, cg, [1.0] * len(dims), 1e-6, True, k)
    Wr, _, vr = gf.gcca_from_moments(G, s, n, dims, k, c=cg, view_weights=[1.0] * len(dims), eps=1e-6, center=True)
    np.testing.assert_allclose(vals, vr[:k], rtol=1e-8, atol=1e-10)
    if _subspace_gap_ok(vr, k):
        for w, r in zip(W, Wr):
            assert col_rel_err(w, r) < 1e-6, ("gcca", dims, cg)


def test_mcca_whitened_operator_lower_bound_is_a_bound():
    """The inequality behind the MCCA hint, checked numerically: eigenvalues of L^-1 (C - blockdiag C) L^-T are
    >= -max_i 1 / (1 - c_i) for R_i = (1 - c_i) C_ii + c_i I = L_i L_i'."""
    rng = np.random.default_rng(5)
    dims = [7, 5, 9]
    X = rng.standard_normal((40, sum(dims))) @ rng.standard_normal((sum(dims), sum(dims)))
    C = np.cov(X, rowvar=False)
    off = np.cumsum([0] + dims)
    for c in ([0.0, 0.0, 0.0], [0.3, 0.9, 0.0], [0.99, 0.5, 0.1]):
        Li = np.zeros_like(C)
        A = C.copy()
        for i in range(3):
            b = slice(off[i], off[i + 1])
            R = (1 - c[i]) * C[b, b] + c[i] * np.eye(dims[i])
            Li[b, b] = np.linalg.inv(np.linalg.cholesky(R))
            A[b, b] = 0.0
        lam = np.linalg.eigvalsh(Li @ A @ Li.T)
        assert lam.min() >= -max(1.0 / (1.0 - ci) for ci in c) - 1e-10
