"""Pin the CPU oracle against golden vectors captured from the real reference.

Both oracle layers (reference-structured and Gram-form) must reproduce the
reference's ``weights_``, ``means_``, ``score`` ... on the stored inputs to
float64 round-off.  CPU only.
"""

import numpy as np
import pytest
import torch

from conftest import col_rel_err, load_golden, rel_err
from oracle import gram_form as gf
from oracle import losses as ol
from oracle import reference_form as rf

TOL = 1e-9   # float64 restatements vs reference (LAPACK path differences only)


def _views(g, prefix):
    out, i = [], 0
    while f"{prefix}{i}" in g:
        out.append(g[f"{prefix}{i}"])
        i += 1
    return out


def _weights(g, tag):
    return _views(g, f"{tag}/w"), _views(g, f"{tag}/mean")


def _check_surface(g, tag, W, means, train, fresh, tol=TOL, wref=None):
    wref = wref if wref is not None else _weights(g, tag)[0]
    for w, r in zip(W, wref):
        assert w.shape == r.shape
        assert col_rel_err(w, r) < tol, tag
    for mu, r in zip(means, _weights(g, tag)[1]):
        np.testing.assert_allclose(mu, r, rtol=1e-6 if r.dtype == np.float32 else 1e-12, atol=1e-12)
    # evaluate the inherited surface with the *golden* weights' signs
    Wa = [w * np.sign(np.sum(w * r, axis=0)) for w, r in zip(W, wref)]
    np.testing.assert_allclose(rf.mean_offdiag_corr(train, Wa, means), g[f"{tag}/score_train"], atol=10 * tol)
    np.testing.assert_allclose(rf.mean_offdiag_corr(fresh, Wa, means), g[f"{tag}/score_fresh"], atol=10 * tol)
    np.testing.assert_allclose(rf.pairwise_corr(train, Wa, means), g[f"{tag}/pairwise_train"], atol=10 * tol)
    for i, t in enumerate(rf.project(train, Wa, means)):
        np.testing.assert_allclose(t[:5], g[f"{tag}/transform{i}"], atol=1e3 * tol)
    for i, l in enumerate(rf.factor_loadings(train, Wa, means)):
        np.testing.assert_allclose(l, g[f"{tag}/loadings{i}"], atol=10 * tol)


RCCA = {
    "cca": dict(c=0.0), "rcca_0.1": dict(c=0.1), "rcca_0.1_0.3": dict(c=[0.1, 0.3]),
    "pls": dict(c=1.0), "rcca_0.1_nocenter": dict(c=0.1, center=False),
}
MCCA = {
    "mcca_c0_pca": dict(c=0.0, pca=True), "mcca_c0_nopca": dict(c=0.0, pca=False),
    "mcca_c0.1_pca": dict(c=0.1, pca=True), "mcca_c0.1_nopca": dict(c=0.1, pca=False),
    "mcca_c0.1_nocenter": dict(c=0.1, center=False),
}
GCCA = {"gcca_c0": dict(c=0.0), "gcca_c0.1": dict(c=0.1), "gcca_c0.1_nocenter": dict(c=0.1, center=False)}


def _cl(c, m):
    return list(c) if isinstance(c, (list, tuple)) else [c] * m


@pytest.mark.parametrize("tag", list(RCCA))
def test_c1_rcca_family(tag):
    g = load_golden("c1_two_view_f64")
    train, fresh = _views(g, "train"), _views(g, "fresh")
    kw = dict(RCCA[tag])
    W, means = rf.rcca_weights(train, 2, **kw)
    _check_surface(g, tag, W, means, train, fresh)
    G, s, n = gf.moments(train)
    W2, means2, _ = gf.rcca_from_moments(G, s, n, [50, 50], 2, c=_cl(kw.get("c", 0.0), 2),
                                         center=kw.get("center", True))
    _check_surface(g, tag, W2, means2, train, fresh)


@pytest.mark.parametrize("tag", list(MCCA))
def test_c1_mcca(tag):
    g = load_golden("c1_two_view_f64")
    train, fresh = _views(g, "train"), _views(g, "fresh")
    kw = dict(MCCA[tag])
    W, means = rf.mcca_weights(train, 2, **kw)
    _check_surface(g, tag, W, means, train, fresh)
    G, s, n = gf.moments(train)
    W2, means2, _ = gf.mcca_from_moments(G, s, n, [50, 50], 2, c=_cl(kw["c"], 2), center=kw.get("center", True))
    _check_surface(g, tag, W2, means2, train, fresh)


@pytest.mark.parametrize("tag", list(GCCA))
def test_c1_gcca(tag):
    g = load_golden("c1_two_view_f64")
    train, fresh = _views(g, "train"), _views(g, "fresh")
    kw = dict(GCCA[tag])
    W, means = rf.gcca_weights(train, 2, **kw)
    _check_surface(g, tag, W, means, train, fresh)
    G, s, n = gf.moments(train)
    W2, means2, _ = gf.gcca_from_moments(G, s, n, [50, 50], 2, c=_cl(kw["c"], 2), center=kw.get("center", True))
    _check_surface(g, tag, W2, means2, train, fresh, tol=1e-8)


def test_three_view_cases():
    g = load_golden("three_view_f64")
    train, fresh = _views(g, "train"), _views(g, "fresh")
    dims = [40, 30, 20]
    G, s, n = gf.moments(train)
    cases = {
        "mcca_c0": ("m", dict(c=0.0)),
        "mcca_c_list": ("m", dict(c=[0.1, 0.2, 0.3], pca=False)),
        "gcca_c0": ("g", dict(c=0.0)),
        "gcca_weighted": ("g", dict(c=0.1, view_weights=[1.0, 1.0, 2.0])),
        "gcca_nocenter": ("g", dict(c=0.2, center=False)),
    }
    for tag, (kind, kw) in cases.items():
        if kind == "m":
            W, means = rf.mcca_weights(train, 3, **kw)
            W2, means2, _ = gf.mcca_from_moments(G, s, n, dims, 3, c=_cl(kw["c"], 3))
        else:
            W, means = rf.gcca_weights(train, 3, **kw)
            W2, means2, _ = gf.gcca_from_moments(G, s, n, dims, 3, c=_cl(kw["c"], 3),
                                                 view_weights=kw.get("view_weights"),
                                                 center=kw.get("center", True))
        _check_surface(g, tag, W, means, train, fresh)
        _check_surface(g, tag, W2, means2, train, fresh, tol=1e-8)


def test_separated_spectrum_and_k_clamp():
    g = load_golden("separated_two_view_f64")
    train, fresh = _views(g, "train"), _views(g, "fresh")
    G, s, n = gf.moments(train)
    dims = [24, 17]
    W, means = rf.rcca_weights(train, 6, c=0.0)
    _check_surface(g, "cca_k6", W, means, train, fresh)
    W, means, sv = gf.rcca_from_moments(G, s, n, dims, 6, c=[0.0, 0.0])
    _check_surface(g, "cca_k6", W, means, train, fresh)
    # for c = 0 the training score equals the singular values of the whitened cross-covariance
    np.testing.assert_allclose(sv, g["cca_k6/score_train"], atol=1e-10)
    W, means, _ = gf.rcca_from_moments(G, s, n, dims, 6, c=[0.2, 0.2])
    _check_surface(g, "rcca_k6_c0.2", W, means, train, fresh)
    W, means, _ = gf.mcca_from_moments(G, s, n, dims, 6, c=[0.05, 0.05])
    _check_surface(g, "mcca_k6_c0.05", W, means, train, fresh)
    W, means, _ = gf.gcca_from_moments(G, s, n, dims, 6, c=[0.05, 0.05])
    _check_surface(g, "gcca_k6_c0.05", W, means, train, fresh, tol=1e-8)
    W, means, _ = gf.rcca_from_moments(G, s, n, dims, 40, c=[0.0, 0.0])
    assert W[0].shape == (24, 17) and W[1].shape == (17, 17)
    _check_surface(g, "cca_k40", W, means, train, fresh, tol=1e-7)


def test_wide_rank_deficient_with_ridge():
    g = load_golden("wide_two_view_f64")
    train, fresh = _views(g, "train"), _views(g, "fresh")
    G, s, n = gf.moments(train)
    dims = [60, 55]
    W, means = rf.rcca_weights(train, 3, c=0.3)
    _check_surface(g, "rcca_c0.3", W, means, train, fresh)
    W, means, _ = gf.rcca_from_moments(G, s, n, dims, 3, c=[0.3, 0.3])
    _check_surface(g, "rcca_c0.3", W, means, train, fresh, tol=1e-8)
    for tag in ("mcca_c0.3", "mcca_c0.3_nopca"):
        W, means, _ = gf.mcca_from_moments(G, s, n, dims, 3, c=[0.3, 0.3])
        _check_surface(g, tag, W, means, train, fresh, tol=1e-8)
    W, means, _ = gf.gcca_from_moments(G, s, n, dims, 3, c=[0.3, 0.3])
    _check_surface(g, "gcca_c0.3", W, means, train, fresh, tol=1e-7)


def test_float32_offset_inputs():
    """fp32 inputs with large means: the reference keeps fp32 for rCCA and
    promotes to fp64 for MCCA/GCCA; the fp64 Gram form must agree to fp32-level."""
    g = load_golden("c1_two_view_f32_offset")
    train = _views(g, "train")
    G, s, n = gf.moments(train)
    W, means, _ = gf.rcca_from_moments(G, s, n, [50, 50], 2, c=[0.1, 0.1])
    for w, r in zip(W, _weights(g, "rcca_0.1")[0]):
        assert r.dtype == np.float32
        assert col_rel_err(w, r) < 1e-3
    W, means, _ = gf.mcca_from_moments(G, s, n, [50, 50], 2, c=[0.1, 0.1])
    for w, r in zip(W, _weights(g, "mcca_c0.1")[0]):
        assert r.dtype == np.float64
        # the reference centres in fp32 (_base.py:97-99) before np.cov promotes, so its
        # own fp64 weights carry fp32 centring error (~1e-4 here); bar = 1e-3 (fp32)
        assert col_rel_err(w, r) < 1e-3
    W, means, _ = gf.gcca_from_moments(G, s, n, [50, 50], 2, c=[0.1, 0.1])
    for w, r in zip(W, _weights(g, "gcca_c0.1")[0]):
        assert col_rel_err(w, r) < 1e-3
    for mu, r in zip(means, _weights(g, "gcca_c0.1")[1]):
        assert r.dtype == np.float32
        np.testing.assert_allclose(mu, r, rtol=1e-6)


def test_joint_data_stream():
    g = load_golden("jointdata_seed0")
    d0, d1 = rf.joint_data(2, 200, 2, [50, 50], 2.0, 0, n_draws=2)
    np.testing.assert_array_equal(d0[0], g["draw0_v0"])
    np.testing.assert_array_equal(d0[1], g["draw0_v1"])
    np.testing.assert_array_equal(d1[0], g["draw1_v0"])
    np.testing.assert_array_equal(d1[1], g["draw1_v1"])


def test_linalg_seams():
    g = load_golden("linalg_seams")
    X = g["X"]
    for c in (0.0, 0.25, 1.0):
        xw, W = rf.thin_svd_whitener(X, c)
        assert col_rel_err(W, g[f"c{c}/W"]) < 1e-10
        W2, lam = gf.whitener_from_gram(X.T @ X, X.shape[0], c)
        assert col_rel_err(W2, g[f"c{c}/W"]) < 1e-9
        xw2 = X @ (W2 * np.sign(np.sum(W2 * g[f"c{c}/W"], axis=0)))
        np.testing.assert_allclose(xw2[:5], g[f"c{c}/X_white_head"], atol=1e-9)
    w, V = rf.top_eigenpairs(g["A"], None, 5)
    np.testing.assert_allclose(w, g["gevp_std/w"], rtol=1e-12)
    assert col_rel_err(V, g["gevp_std/V"]) < 1e-10
    w, V = rf.top_eigenpairs(g["A"], g["B"], 5)
    np.testing.assert_allclose(w, g["gevp_gen/w"], rtol=1e-11)
    assert col_rel_err(V, g["gevp_gen/V"]) < 1e-10


def _loss_tags(g):
    return sorted({k.rsplit("/", 1)[0] for k in g if k.startswith("cca/")})


def test_cca_loss_value_and_grad():
    g = load_golden("losses")
    for tag in _loss_tags(g):
        z1, z2 = g[tag + "/z1"], g[tag + "/z2"]
        eps = 1e-5 if "unequal" in tag else float(tag.split("eps")[1])
        f32 = z1.dtype == np.float32
        # reference-structured torch restatement: same dtype as the inputs
        t1 = torch.tensor(z1, requires_grad=True)
        t2 = torch.tensor(z2, requires_grad=True)
        loss = ol.cca_loss_autograd(t1, t2, eps)
        loss.backward()
        tol = 2e-3 if f32 else 1e-10
        assert abs(loss.item() - g[tag + "/loss"]) <= tol * abs(g[tag + "/loss"]), tag
        assert rel_err(t1.grad.numpy(), g[tag + "/g1"]) < (5e-2 if f32 else 1e-8), tag
        # closed form (float64 spec of the HIP path)
        l, g1, g2 = ol.cca_loss_closed_form(z1, z2, eps)
        assert abs(l - g[tag + "/loss"]) <= (1e-3 if f32 else 1e-9) * abs(g[tag + "/loss"]), tag
        assert rel_err(g1, g[tag + "/g1"]) < (5e-2 if f32 else 1e-7), tag
        assert rel_err(g2, g[tag + "/g2"]) < (5e-2 if f32 else 1e-7), tag


def test_mcca_loss_and_inv_sqrtm():
    g = load_golden("losses")
    zs = [g[f"mcca/z{i}"] for i in range(3)]
    l, grads = ol.mcca_loss_closed_form(zs, 1e-5)
    assert abs(l - g["mcca/loss"]) < 1e-6 * abs(g["mcca/loss"])   # reference accumulates in fp32
    for i in range(3):
        assert rel_err(grads[i], g[f"mcca/g{i}"]) < 1e-8
    A = torch.tensor(g["inv_sqrtm/A"])
    np.testing.assert_allclose(ol.inv_sqrtm_eigh(A, 1e-5).numpy(), g["inv_sqrtm/out_eps1e-5"], atol=1e-9)
    np.testing.assert_allclose(ol.inv_sqrtm_eigh(A, 0.5).numpy(), g["inv_sqrtm/out_eps0.5"], atol=1e-10)


def test_gcca_gram_form_topk_hook_matches_the_dense_solve():
    """``oracle.gram_form.gcca_from_moments(topk=...)`` (used by the configs[4]-sized GPU test with a Lanczos solver) gives the dense
    ``eigh`` result: weights, eigenvalues, non-uniform view weights and per-view ridge."""
    import scipy.sparse.linalg as spla

    from oracle import gram_form as gf
    from oracle import reference_form as rf

    views = rf.joint_data(3, 400, 4, [30, 24, 20], 2.0, 1)
    G, s, n = gf.moments(views)
    kw = dict(c=[0.1, 0.2, 0.05], view_weights=[1.0, 2.0, 0.5])
    W0, _, l0 = gf.gcca_from_moments(G, s, n, [30, 24, 20], 4, **kw)

    def lanczos(K, k):
        lam, U = spla.eigsh(K, k=k, which="LA", tol=1e-13)
        o = np.argsort(lam)[::-1]
        return lam[o], U[:, o]

    W1, _, l1 = gf.gcca_from_moments(G, s, n, [30, 24, 20], 4, topk=lanczos, **kw)
    np.testing.assert_allclose(l1, l0, rtol=1e-11)
    for a, b in zip(W0, W1):
        sgn = np.sign(np.sum(a * b, axis=0))
        np.testing.assert_allclose(b * sgn, a, atol=1e-10 * np.abs(a).max())
