"""GPU parity: the estimator surface against golden vectors captured from the reference.

Bars (BASELINE.json north_star): weights and ``score`` within 1e-5 relative for float64
inputs and 1e-3 for float32 inputs, sign-aligned per column.
"""

import numpy as np
import pytest

from conftest import col_rel_err, load_golden

pytestmark = pytest.mark.gpu

TOL64 = 1e-5
TOL32 = 1e-3


def _views(g, prefix):
    out, i = [], 0
    while f"{prefix}{i}" in g:
        out.append(g[f"{prefix}{i}"])
        i += 1
    return out


def _align(model, g, tag):
    """Flip fitted columns to the golden signs so that transform/loadings compare directly."""
    for i in range(len(model.weights_)):
        r = g[f"{tag}/w{i}"]
        s = np.sign(np.sum(model.weights_[i] * r, axis=0))
        s[s == 0] = 1
        model.weights_[i] = model.weights_[i] * s.astype(model.weights_[i].dtype)


def _check(model, g, tag, train, fresh, tol):
    for i, w in enumerate(model.weights_):
        r = g[f"{tag}/w{i}"]
        assert w.shape == r.shape and w.dtype == r.dtype, (tag, w.dtype, r.dtype)
        assert col_rel_err(w, r) < tol, (tag, i, col_rel_err(w, r))
    for i, m in enumerate(model.means_):
        r = g[f"{tag}/mean{i}"]
        assert m.dtype == r.dtype
        np.testing.assert_allclose(m, r, rtol=1e-5 if r.dtype == np.float32 else 1e-11, atol=1e-6 if r.dtype == np.float32 else 1e-12)
    _align(model, g, tag)
    np.testing.assert_allclose(model.score(train), g[f"{tag}/score_train"], atol=tol)
    np.testing.assert_allclose(model.score(fresh), g[f"{tag}/score_fresh"], atol=10 * tol)
    np.testing.assert_allclose(model.pairwise_correlations(train), g[f"{tag}/pairwise_train"], atol=10 * tol)
    for i, t in enumerate(model.transform(train)):
        np.testing.assert_allclose(t[:5], g[f"{tag}/transform{i}"], rtol=50 * tol, atol=50 * tol)
    for i, l in enumerate(model.get_factor_loadings(train)):
        np.testing.assert_allclose(l, g[f"{tag}/loadings{i}"], atol=20 * tol)


def _specs():
    from cca_zoo_amd.linear import CCA, GCCA, MCCA, PLS, rCCA

    return {
        "cca": lambda: CCA(latent_dimensions=2),
        "rcca_0.1": lambda: rCCA(latent_dimensions=2, c=0.1),
        "rcca_0.1_0.3": lambda: rCCA(latent_dimensions=2, c=[0.1, 0.3]),
        "pls": lambda: PLS(latent_dimensions=2),
        "rcca_0.1_nocenter": lambda: rCCA(latent_dimensions=2, c=0.1, center=False),
        "mcca_c0_pca": lambda: MCCA(latent_dimensions=2, c=0.0, pca=True),
        "mcca_c0_nopca": lambda: MCCA(latent_dimensions=2, c=0.0, pca=False),
        "mcca_c0.1_pca": lambda: MCCA(latent_dimensions=2, c=0.1, pca=True),
        "mcca_c0.1_nopca": lambda: MCCA(latent_dimensions=2, c=0.1, pca=False),
        "mcca_c0.1_nocenter": lambda: MCCA(latent_dimensions=2, c=0.1, center=False),
        "gcca_c0": lambda: GCCA(latent_dimensions=2, c=0.0),
        "gcca_c0.1": lambda: GCCA(latent_dimensions=2, c=0.1),
        "gcca_c0.1_nocenter": lambda: GCCA(latent_dimensions=2, c=0.1, center=False),
    }


@pytest.mark.parametrize("tag", ["cca", "rcca_0.1", "rcca_0.1_0.3", "pls", "rcca_0.1_nocenter", "mcca_c0_pca",
                                 "mcca_c0_nopca", "mcca_c0.1_pca", "mcca_c0.1_nopca", "mcca_c0.1_nocenter",
                                 "gcca_c0", "gcca_c0.1", "gcca_c0.1_nocenter"])
def test_c1_configuration(tag):
    """BASELINE configs[0]: CCA(latent_dimensions=2) on JointData n=200, 2 x 50 (+ its siblings)."""
    g = load_golden("c1_two_view_f64")
    train, fresh = _views(g, "train"), _views(g, "fresh")
    model = _specs()[tag]().fit(train)
    assert model.n_views_ == 2 and model.n_features_in_ == [50, 50] and model.n_samples_ == 200
    _check(model, g, tag, train, fresh, TOL64)


def test_three_views():
    from cca_zoo_amd.linear import GCCA, MCCA

    g = load_golden("three_view_f64")
    train, fresh = _views(g, "train"), _views(g, "fresh")
    cases = {
        "mcca_c0": MCCA(latent_dimensions=3, c=0.0),
        "mcca_c_list": MCCA(latent_dimensions=3, c=[0.1, 0.2, 0.3], pca=False),
        "gcca_c0": GCCA(latent_dimensions=3, c=0.0),
        "gcca_weighted": GCCA(latent_dimensions=3, c=0.1, view_weights=[1.0, 1.0, 2.0]),
        "gcca_nocenter": GCCA(latent_dimensions=3, c=0.2, center=False),
    }
    for tag, model in cases.items():
        _check(model.fit(train), g, tag, train, fresh, TOL64)


def test_separated_spectrum_k_clamp_and_wide():
    from cca_zoo_amd.linear import CCA, GCCA, MCCA, rCCA

    g = load_golden("separated_two_view_f64")
    train, fresh = _views(g, "train"), _views(g, "fresh")
    _check(CCA(latent_dimensions=6).fit(train), g, "cca_k6", train, fresh, TOL64)
    _check(rCCA(latent_dimensions=6, c=0.2).fit(train), g, "rcca_k6_c0.2", train, fresh, TOL64)
    _check(MCCA(latent_dimensions=6, c=0.05).fit(train), g, "mcca_k6_c0.05", train, fresh, TOL64)
    _check(GCCA(latent_dimensions=6, c=0.05).fit(train), g, "gcca_k6_c0.05", train, fresh, TOL64)
    m = CCA(latent_dimensions=40).fit(train)          # k clamps to min(d1, d2) = 17
    assert m.weights_[0].shape == (24, 17)
    for i in range(2):
        assert col_rel_err(m.weights_[i][:, :12], g[f"cca_k40/w{i}"][:, :12]) < 1e-4
    g = load_golden("wide_two_view_f64")               # d > n with ridge
    train, fresh = _views(g, "train"), _views(g, "fresh")
    _check(rCCA(latent_dimensions=3, c=0.3).fit(train), g, "rcca_c0.3", train, fresh, TOL64)
    _check(MCCA(latent_dimensions=3, c=0.3).fit(train), g, "mcca_c0.3", train, fresh, TOL64)
    _check(MCCA(latent_dimensions=3, c=0.3, pca=False).fit(train), g, "mcca_c0.3_nopca", train, fresh, TOL64)
    _check(GCCA(latent_dimensions=3, c=0.3).fit(train), g, "gcca_c0.3", train, fresh, TOL64)


def test_float32_inputs_dtype_flow_and_bar():
    """fp32 views with large means: rCCA returns fp32 weights, MCCA/GCCA float64; bar 1e-3."""
    from cca_zoo_amd.linear import CCA, GCCA, MCCA, rCCA

    g = load_golden("c1_two_view_f32_offset")
    train, fresh = _views(g, "train"), _views(g, "fresh")
    assert train[0].dtype == np.float32
    _check(rCCA(latent_dimensions=2, c=0.1).fit(train), g, "rcca_0.1", train, fresh, TOL32)
    _check(CCA(latent_dimensions=2).fit(train), g, "cca", train, fresh, TOL32)
    _check(MCCA(latent_dimensions=2, c=0.1).fit(train), g, "mcca_c0.1", train, fresh, TOL32)
    _check(GCCA(latent_dimensions=2, c=0.1).fit(train), g, "gcca_c0.1", train, fresh, TOL32)


def test_reference_invariants():
    """The property tests the reference itself pins (tests/linear/test_eigendecomposition.py)."""
    from cca_zoo_amd.linear import CCA, GCCA, MCCA, PLS, rCCA

    rng = np.random.default_rng(0)
    X1, X2 = rng.standard_normal((50, 10)), rng.standard_normal((50, 8))
    # identical views -> score 1 (:330-336)
    np.testing.assert_allclose(CCA(latent_dimensions=3).fit([X1, X1]).score([X1, X1]), 1.0, atol=1e-6)
    # correlations non-increasing (:339-342); rCCA(c=0) == CCA (:345-353); MCCA 2 views == CCA (:411-416)
    sc = CCA(latent_dimensions=4).fit([X1, X2]).score([X1, X2])
    assert np.all(np.diff(sc) <= 1e-9)
    np.testing.assert_allclose(rCCA(latent_dimensions=4, c=0.0).fit([X1, X2]).score([X1, X2]), sc, atol=1e-6)
    np.testing.assert_allclose(MCCA(latent_dimensions=4).fit([X1, X2]).score([X1, X2]), sc, atol=1e-6)
    # canonical variates uncorrelated (:419-429)
    z1, z2 = CCA(latent_dimensions=3).fit([X1, X2]).transform([X1, X2])
    c = np.corrcoef(z1, rowvar=False)
    np.testing.assert_allclose(c - np.diag(np.diag(c)), 0.0, atol=1e-6)
    # fit_transform == fit().transform() (:112-137)
    for cls in (CCA, PLS, MCCA, GCCA):
        a = cls(latent_dimensions=2).fit_transform([X1, X2])
        b = cls(latent_dimensions=2).fit([X1, X2]).transform([X1, X2])
        for u, v in zip(a, b):
            np.testing.assert_allclose(np.abs(u), np.abs(v), atol=1e-9)
    # correlated views score high (:321-327)
    z = rng.standard_normal((200, 2))
    A = z @ rng.standard_normal((2, 10)) + 0.1 * rng.standard_normal((200, 10))
    B = z @ rng.standard_normal((2, 8)) + 0.1 * rng.standard_normal((200, 8))
    assert CCA(latent_dimensions=2).fit([A, B]).score([A, B]).min() > 0.95
    # error behaviour
    with pytest.raises(ValueError, match="exactly 2 views"):
        CCA().fit([X1, X2, X1])
    with pytest.raises(ValueError, match="At least 2 views"):
        MCCA().fit([X1])
    with pytest.raises(ValueError, match="same number of samples"):
        MCCA().fit([X1, X2[:40]])


def test_device_resident_views_match_host_views():
    """torch CUDA tensors in -> same fit as host arrays (and transform stays on the device)."""
    import torch

    from cca_zoo_amd.linear import MCCA, rCCA

    rng = np.random.default_rng(4)
    z = rng.standard_normal((3000, 4)) * np.linspace(2, 0.5, 4)
    views = [(z @ rng.standard_normal((4, d)) + rng.standard_normal((3000, d))).astype(np.float32) for d in (256, 256)]
    tv = [torch.as_tensor(v, device="cuda") for v in views]
    a = rCCA(latent_dimensions=4, c=0.1).fit(views)
    b = rCCA(latent_dimensions=4, c=0.1).fit(tv)
    # (host-streamed chunks always take the pilot-shifted K1, HBM-resident centred data the plain one: two fp32
    # accumulation orders of the same moments -- the float32 bar against the reference is 1e-3)
    for u, v in zip(a.weights_, b.weights_):
        assert col_rel_err(u, v) < 1e-4
    np.testing.assert_allclose(a.score(views), b.score(tv), atol=1e-5)
    out = b.transform(tv)
    assert out[0].is_cuda and out[0].shape == (3000, 4)
    m = MCCA(latent_dimensions=3, c=0.1).fit(tv)
    assert m.weights_[0].dtype == np.float64


def test_config2_shape_reduced_rows_fp32_bar():
    """BASELINE configs[1] shape (2 x 1024, k=32, fp32) at n=20000 against the fp64 oracle."""
    from cca_zoo_amd.datasets import JointData
    from cca_zoo_amd.linear import rCCA
    from oracle import gram_form as gf

    jd = JointData(n_views=2, n_samples=20000, n_features=[1024, 1024], latent_dimensions=32,
                   random_state=0, latent_scales=list(np.linspace(2.0, 0.5, 32)))
    views = [v.astype(np.float32) for v in jd.sample()]
    model = rCCA(latent_dimensions=32, c=0.1).fit(views)
    G, s, n = gf.moments(views)
    W, _, _ = gf.rcca_from_moments(G, s, n, [1024, 1024], 32, c=[0.1, 0.1])
    for u, r in zip(model.weights_, W):
        assert u.dtype == np.float32
        assert col_rel_err(u, r) < TOL32


def test_config2_full_size_properties():
    """BASELINE configs[1] at FULL size (rCCA n=100k, 2 x 1024, k=32, float32): size-independent
    properties instead of an O(n d^2) oracle -- normalisation w'R w = 1, uncorrelated variates,
    training score == singular values for c=0, shard additivity of the moments, CCA == MCCA scores."""
    import torch

    from cca_zoo_amd import _backend
    from cca_zoo_amd._moments import compute_moments
    from cca_zoo_amd.datasets import JointData
    from cca_zoo_amd.linear import CCA, MCCA, rCCA

    n, d, k = 100_000, 1024, 32
    jd = JointData(n_views=2, n_samples=n, n_features=[d, d], latent_dimensions=k, random_state=3,
                   latent_scales=list(np.linspace(2.0, 0.5, k)))
    views = jd.sample_device(device="cuda", dtype=torch.float32, n_samples=n, seed=3)
    h = _backend.default_handle(0)
    # shard additivity (the multi-GPU identity): moments(all rows) == moments(shard A) + moments(shard B)
    mom, keep, nt, dims, kind = compute_moments(views, h)
    full = h.to_host(mom, (2 * d * 2 * d + 2 * d,))
    ma, ka, _, _, _ = compute_moments([v[:37_001] for v in views], h)
    mb, kb, _, _, _ = compute_moments([v[37_001:] for v in views], h)
    parts = h.to_host(ma, full.shape) + h.to_host(mb, full.shape)
    scale = np.abs(full).max()
    # fp32 MFMA accumulation inside a <= 16384-row chunk: ~4e-6 of an entry per chunk (the two shardings chunk the rows
    # differently), fp64 across chunks
    assert np.max(np.abs(parts - full)) < 6e-6 * scale
    # rCCA(c=0.1): every weight column satisfies w' ((1-c) C_ii + c I) w = 1
    m = rCCA(latent_dimensions=k, c=0.1).fit(views)
    assert m.weights_[0].dtype == np.float32 and m.weights_[0].shape == (d, k)
    X = views[0].double()
    C11 = torch.cov(X.T).cpu().numpy()
    W1 = m.weights_[0].astype(np.float64)
    np.testing.assert_allclose(np.diag(W1.T @ (0.9 * C11 + 0.1 * np.eye(d)) @ W1), 1.0, atol=2e-3)
    # CCA: variates uncorrelated within a view, unit variance, train score == singular values, MCCA agrees
    cca = CCA(latent_dimensions=k).fit(views)
    z1, z2 = cca.transform(views)
    c = torch.cov(z1.double().T).cpu().numpy()
    np.testing.assert_allclose(c, np.eye(k), atol=2e-3)
    sc = cca.score(views)
    np.testing.assert_allclose(sc, cca.singular_values_, atol=1e-3)
    assert np.all(np.diff(sc) <= 1e-6) and sc[0] > 0.9
    np.testing.assert_allclose(MCCA(latent_dimensions=k).fit(views).score(views), sc, atol=1e-3)
