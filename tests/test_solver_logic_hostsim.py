"""Driver logic of libccz (cca_zoo_amd/csrc/solve.cpp) on the host test double.

The product C++ solver drivers are compiled against host loops
(tests/hostsim/ops_host.cpp) and called through the same ctypes signatures the
GPU build uses.  Checks them against the goldens captured from the reference and
against NumPy.  No GPU; nothing here is a product path.
"""

import ctypes as C

import numpy as np
import pytest

from conftest import col_rel_err, load_golden
from hostsim_util import hostsim_handle, pack_moments
from oracle import gram_form as gf


@pytest.fixture(scope="module")
def H():
    return hostsim_handle()


def _views(g, prefix):
    out, i = [], 0
    while f"{prefix}{i}" in g:
        out.append(g[f"{prefix}{i}"])
        i += 1
    return out


def _gold_w(g, tag):
    return _views(g, f"{tag}/w"), _views(g, f"{tag}/mean")


def _mom(views):
    G, s, n = gf.moments(views)
    return pack_moments(G, s), n


C1_RCCA = {"cca": ([0.0, 0.0], True), "rcca_0.1": ([0.1, 0.1], True), "rcca_0.1_0.3": ([0.1, 0.3], True),
           "pls": ([1.0, 1.0], True), "rcca_0.1_nocenter": ([0.1, 0.1], False)}


@pytest.mark.parametrize("tag", list(C1_RCCA))
def test_rcca_solve_c1(H, tag):
    g = load_golden("c1_two_view_f64")
    train = _views(g, "train")
    mom, n = _mom(train)
    c, center = C1_RCCA[tag]
    W, means, vals = H.rcca_solve(mom, n, [50, 50], c, center, 2)
    Wr, mr = _gold_w(g, tag)
    for w, r in zip(W, Wr):
        assert col_rel_err(w, r) < 1e-8
    for a, b in zip(means, mr):
        np.testing.assert_allclose(a, b, atol=1e-12)
    if tag == "cca":
        np.testing.assert_allclose(vals, g["cca/score_train"], atol=1e-9)


@pytest.mark.parametrize("tag,c,center", [
    ("mcca_c0_pca", 0.0, True), ("mcca_c0_nopca", 0.0, True), ("mcca_c0.1_pca", 0.1, True),
    ("mcca_c0.1_nopca", 0.1, True), ("mcca_c0.1_nocenter", 0.1, False)])
def test_mcca_solve_c1(H, tag, c, center):
    g = load_golden("c1_two_view_f64")
    mom, n = _mom(_views(g, "train"))
    W, means, vals = H.mcca_solve(mom, n, [50, 50], [c, c], 1e-6, center, 2)
    Wr, mr = _gold_w(g, tag)
    for w, r in zip(W, Wr):
        assert col_rel_err(w, r) < 1e-8
    for a, b in zip(means, mr):
        np.testing.assert_allclose(a, b, atol=1e-12)


@pytest.mark.parametrize("tag,c,center", [("gcca_c0", 0.0, True), ("gcca_c0.1", 0.1, True),
                                          ("gcca_c0.1_nocenter", 0.1, False)])
def test_gcca_solve_c1(H, tag, c, center):
    g = load_golden("c1_two_view_f64")
    mom, n = _mom(_views(g, "train"))
    W, means, vals = H.gcca_solve(mom, n, [50, 50], [c, c], [1.0, 1.0], 1e-6, center, 2)
    Wr, mr = _gold_w(g, tag)
    for w, r in zip(W, Wr):
        assert col_rel_err(w, r) < 1e-7
    for a, b in zip(means, mr):
        np.testing.assert_allclose(a, b, atol=1e-12)


def test_three_view(H):
    g = load_golden("three_view_f64")
    mom, n = _mom(_views(g, "train"))
    dims = [40, 30, 20]
    W, _, _ = H.mcca_solve(mom, n, dims, [0.0] * 3, 1e-6, True, 3)
    for w, r in zip(W, _gold_w(g, "mcca_c0")[0]):
        assert col_rel_err(w, r) < 1e-8
    W, _, _ = H.mcca_solve(mom, n, dims, [0.1, 0.2, 0.3], 1e-6, True, 3)
    for w, r in zip(W, _gold_w(g, "mcca_c_list")[0]):
        assert col_rel_err(w, r) < 1e-8
    W, _, _ = H.gcca_solve(mom, n, dims, [0.0] * 3, [1.0] * 3, 1e-6, True, 3)
    for w, r in zip(W, _gold_w(g, "gcca_c0")[0]):
        assert col_rel_err(w, r) < 1e-7
    W, _, _ = H.gcca_solve(mom, n, dims, [0.1] * 3, [1.0, 1.0, 2.0], 1e-6, True, 3)
    for w, r in zip(W, _gold_w(g, "gcca_weighted")[0]):
        assert col_rel_err(w, r) < 1e-7
    W, means, _ = H.gcca_solve(mom, n, dims, [0.2] * 3, [1.0] * 3, 1e-6, False, 3)
    for w, r in zip(W, _gold_w(g, "gcca_nocenter")[0]):
        assert col_rel_err(w, r) < 1e-7
    assert all(np.all(mu == 0) for mu in means)


def test_separated_and_clamp(H):
    g = load_golden("separated_two_view_f64")
    mom, n = _mom(_views(g, "train"))
    W, _, vals = H.rcca_solve(mom, n, [24, 17], [0.0, 0.0], True, 6)
    for w, r in zip(W, _gold_w(g, "cca_k6")[0]):
        assert col_rel_err(w, r) < 1e-8
    np.testing.assert_allclose(vals, g["cca_k6/score_train"], atol=1e-10)
    W, _, vals = H.rcca_solve(mom, n, [24, 17], [0.0, 0.0], True, 40)
    assert W[0].shape == (24, 17) and W[1].shape == (17, 17) and vals.shape == (17,)
    for w, r in zip(W, _gold_w(g, "cca_k40")[0]):
        assert col_rel_err(w, r) < 1e-6
    W, _, _ = H.gcca_solve(mom, n, [24, 17], [0.05, 0.05], [1.0, 1.0], 1e-6, True, 6)
    for w, r in zip(W, _gold_w(g, "gcca_k6_c0.05")[0]):
        assert col_rel_err(w, r) < 1e-7


def test_wide_rank_deficient(H):
    g = load_golden("wide_two_view_f64")
    mom, n = _mom(_views(g, "train"))
    W, _, _ = H.rcca_solve(mom, n, [60, 55], [0.3, 0.3], True, 3)
    for w, r in zip(W, _gold_w(g, "rcca_c0.3")[0]):
        assert col_rel_err(w, r) < 1e-7
    W, _, _ = H.mcca_solve(mom, n, [60, 55], [0.3, 0.3], 1e-6, True, 3)
    for w, r in zip(W, _gold_w(g, "mcca_c0.3")[0]):
        assert col_rel_err(w, r) < 1e-7
    # GCCA with d > n needs the eigen pseudo-inverse branch (Gx_ii singular)
    W, _, _ = H.gcca_solve(mom, n, [60, 55], [0.3, 0.3], [1.0, 1.0], 1e-6, True, 3)
    for w, r in zip(W, _gold_w(g, "gcca_c0.3")[0]):
        assert col_rel_err(w, r) < 1e-6


def test_rank_deficient_c0_falls_back_to_floor(H):
    """c = 0 on d > n data: Cholesky fails, the eigen-floored whitener is used;
    result must satisfy the CCA invariants (training correlations ~ 1)."""
    g = load_golden("wide_two_view_f64")
    train = _views(g, "train")
    mom, n = _mom(train)
    W, means, vals = H.rcca_solve(mom, n, [60, 55], [0.0, 0.0], True, 3)
    assert np.all(np.isfinite(W[0])) and np.all(np.isfinite(W[1]))
    np.testing.assert_allclose(vals, 1.0, atol=1e-6)


def test_mcca_eps_shift_branch(H):
    """c = 0 with a duplicated feature -> min eigenvalue 0 < eps -> B is shifted by eps - min_eig."""
    rng = np.random.default_rng(5)
    z = rng.standard_normal((150, 2))
    v1 = z @ rng.standard_normal((2, 6)) + 0.3 * rng.standard_normal((150, 6))
    v1[:, 5] = v1[:, 4]                                    # exactly collinear
    v2 = z @ rng.standard_normal((2, 5)) + 0.3 * rng.standard_normal((150, 5))
    from oracle import reference_form as rf
    Wr, _ = rf.mcca_weights([v1, v2], 2, c=0.0, pca=False, eps=1e-3)
    mom, n = _mom([v1, v2])
    W, _, _ = H.mcca_solve(mom, n, [6, 5], [0.0, 0.0], 1e-3, True, 2)
    for w, r in zip(W, Wr):
        assert col_rel_err(w, r) < 1e-6
    Wg_r, _ = rf.gcca_weights([v1, v2], 2, c=0.0, eps=1e-3)
    Wg, _, _ = H.gcca_solve(mom, n, [6, 5], [0.0, 0.0], [1.0, 1.0], 1e-3, True, 2)
    # the reference's pinv on an exactly collinear view is itself rank-revealing; compare projections
    for v, w, r in zip([v1, v2], Wg, Wg_r):
        a = (v - v.mean(0)) @ w
        b = (v - v.mean(0)) @ r
        assert col_rel_err(a, b) < 1e-5


def test_error_codes(H):
    mom = np.zeros(4 * 4 + 4)
    with pytest.raises(ValueError, match="ridge"):
        H.rcca_solve(mom, 10, [2, 2], [1.5, 0.0], True, 1)
    with pytest.raises(ValueError, match="samples"):
        H.rcca_solve(mom, 1, [2, 2], [0.1, 0.1], True, 1)
    with pytest.raises(ValueError, match="latent"):
        H.rcca_solve(mom, 10, [2, 2], [0.1, 0.1], True, 0)
    with pytest.raises(ValueError, match="eps"):
        H.mcca_solve(mom, 10, [2, 2], [0.1, 0.1], 0.0, True, 1)
    with pytest.raises(np.linalg.LinAlgError):     # all-zero data, c = 0: nothing to whiten
        H.rcca_solve(mom, 10, [2, 2], [0.0, 0.0], True, 1)


# ---- dense seams -----------------------------------------------------------------------
def _call(H, name, *args):
    H.check(getattr(H.lib, name)(H.raw, *args))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_syevj_indefinite_with_plus_minus_pairs(H):
    rng = np.random.default_rng(0)
    T = rng.standard_normal((7, 5))
    A = np.block([[np.zeros((7, 7)), T], [T.T, np.zeros((5, 5))]])     # eigenvalues come in +/- pairs
    A0 = A.copy()
    w, V = np.zeros(12), np.zeros((12, 12))
    sw = C.c_int(0)
    _call(H, "ccz_syevj", _p(A), 12, _p(w), _p(V), C.byref(sw))
    np.testing.assert_allclose(w, np.linalg.eigvalsh(A0)[::-1], atol=1e-12)
    np.testing.assert_allclose(V @ A0 @ V.T, np.diag(w), atol=1e-11)
    np.testing.assert_allclose(V @ V.T, np.eye(12), atol=1e-13)
    assert 1 <= sw.value <= 30


@pytest.mark.parametrize("d", [2, 3, 79, 80, 159, 160, 161])
def test_syevj_two_sided_sizes_and_the_one_sided_path_beyond(H, d):
    """The Rayleigh-Ritz eigen-solve: two-sided tournament Jacobi up to d = 160 (odd sizes pad a zero row / column),
    the one-sided row Jacobi beyond -- both behind ccz_syevj; the host double runs the device kernel's formulation."""
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d))
    A = A + A.T
    A0 = A.copy()
    w, V = np.zeros(d), np.zeros((d, d))
    sw = C.c_int(0)
    _call(H, "ccz_syevj", _p(A), d, _p(w), _p(V), C.byref(sw))
    scale = np.abs(A0).sum(1).max()
    np.testing.assert_allclose(w, np.linalg.eigvalsh(A0)[::-1], atol=1e-13 * scale)
    np.testing.assert_allclose(V @ A0 @ V.T, np.diag(w), atol=1e-12 * scale)
    np.testing.assert_allclose(V @ V.T, np.eye(d), atol=1e-12)
    assert 1 <= sw.value <= 30


@pytest.mark.parametrize("kind,d", [("sym", 200), ("cov", 256), ("lowrank", 192), ("pm", 230)])
def test_syevj_block_jacobi_structure_on_the_host_double(H, kind, d):
    """Above d = 160 ccz_syevj runs the BLOCKED Jacobi (csrc/evd_block.hip); the host double restates its structure --
    32-wide blocks, tournament over the blocks, cross-block rotation rounds (a full tournament in the first round of every
    sweep), zero padding to a multiple of 64 -- so its convergence and bookkeeping are tested here without a GPU:
    Wigner, graded covariance, rank-deficient (a 128-fold eigenvalue 0) and +/- paired spectra."""
    rng = np.random.default_rng(d)
    if kind == "sym":
        A = rng.standard_normal((d, d)); A = A + A.T
    elif kind == "cov":
        X = rng.standard_normal((3 * d, d)) * np.linspace(2.0, 0.05, d); A = X.T @ X / (3 * d - 1)
    elif kind == "lowrank":
        X = rng.standard_normal((d // 3, d)); A = X.T @ X
    else:
        T = rng.standard_normal((d // 2, d - d // 2))
        A = np.block([[np.zeros((d // 2, d // 2)), T], [T.T, np.zeros((d - d // 2, d - d // 2))]])
    A0 = A.copy()
    w, V = np.zeros(d), np.zeros((d, d))
    sw = C.c_int(0)
    _call(H, "ccz_syevj", _p(A), d, _p(w), _p(V), C.byref(sw))
    np.testing.assert_array_equal(A, A0)                                            # the input is only read
    nrm = np.abs(np.linalg.eigvalsh(A0)).max()
    np.testing.assert_allclose(w, np.linalg.eigvalsh(A0)[::-1], atol=1e-12 * nrm)
    assert np.linalg.norm(A0 @ V.T - V.T * w) < 1e-11 * np.linalg.norm(A0)
    assert np.linalg.norm(V @ V.T - np.eye(d)) < 1e-11
    assert 1 <= sw.value <= 40


def test_syevj_rejects_non_finite_input(H):
    A = np.eye(40)
    A[3, 7] = A[7, 3] = np.inf
    w, V = np.zeros(40), np.zeros((40, 40))
    with pytest.raises(ValueError, match="non-finite"):
        _call(H, "ccz_syevj", _p(A), 40, _p(w), _p(V), None)


@pytest.mark.parametrize("shape", [(9, 14), (14, 9), (6, 6)])
def test_gesvj(H, shape):
    rng = np.random.default_rng(1)
    A = rng.standard_normal(shape)
    p, q = shape
    r = min(p, q)
    U, s, Vt = np.zeros((p, r)), np.zeros(r), np.zeros((r, q))
    _call(H, "ccz_gesvj", _p(A), p, q, _p(U), _p(s), _p(Vt), None)
    np.testing.assert_allclose(s, np.linalg.svd(A, compute_uv=False), atol=1e-12)
    np.testing.assert_allclose(U * s @ Vt, A, atol=1e-12)


@pytest.mark.parametrize("p,k", [(60, 5), (300, 12), (260, 40)])
def test_gevp_topk_standard_and_generalised(H, p, k):
    """p <= 192 takes the direct Jacobi path; p = 300 the Chebyshev subspace iteration."""
    rng = np.random.default_rng(2)
    lam = np.concatenate([np.linspace(5.0, 2.0, k), rng.uniform(-3.0, 1.0, p - k)])
    Qm, _ = np.linalg.qr(rng.standard_normal((p, p)))
    A = (Qm * lam) @ Qm.T
    A = 0.5 * (A + A.T)
    w, V = np.zeros(k), np.zeros((p, k))
    _call(H, "ccz_gevp_topk", _p(A), None, p, k, _p(w), _p(V))
    np.testing.assert_allclose(w, np.sort(lam)[::-1][:k], atol=1e-9)
    assert np.linalg.norm(A @ V - V * w) < 1e-8 * np.linalg.norm(A)
    Bm = rng.standard_normal((p, p))
    Bm = Bm @ Bm.T / p + np.eye(p)
    import scipy.linalg
    wr, Vr = scipy.linalg.eigh(A, Bm, subset_by_index=[p - k, p - 1])
    _call(H, "ccz_gevp_topk", _p(A), _p(Bm), p, k, _p(w), _p(V))
    np.testing.assert_allclose(w, wr[::-1], atol=1e-9)
    assert col_rel_err(V, Vr[:, ::-1]) < 1e-6
    np.testing.assert_allclose(np.diag(V.T @ Bm @ V), 1.0, atol=1e-9)


@pytest.mark.parametrize("p,q,k", [(40, 30, 4), (250, 320, 10), (320, 250, 10)])
def test_svd_topk(H, p, q, k):
    rng = np.random.default_rng(3)
    r = min(p, q)
    sv = np.concatenate([np.linspace(3.0, 1.5, k), rng.uniform(0.0, 1.0, r - k)])
    U0, _ = np.linalg.qr(rng.standard_normal((p, r)))
    V0, _ = np.linalg.qr(rng.standard_normal((q, r)))
    T = (U0 * sv) @ V0.T
    U, s, V = np.zeros((p, k)), np.zeros(k), np.zeros((q, k))
    _call(H, "ccz_svd_topk", _p(T), p, q, k, _p(U), _p(s), _p(V))
    np.testing.assert_allclose(s, np.sort(sv)[::-1][:k], atol=1e-9)
    np.testing.assert_allclose(T @ V, U * s, atol=1e-8)
    np.testing.assert_allclose(U.T @ U, np.eye(k), atol=1e-9)


def test_whitener_and_inv_sqrtm(H):
    g = load_golden("linalg_seams")
    X = g["X"]
    n, d = X.shape
    Gxx = np.ascontiguousarray(X.T @ X)
    for c in (0.0, 0.25, 1.0):
        W, lam, r = np.zeros((d, d)), np.zeros(d), C.c_int64(0)
        _call(H, "ccz_whitener", _p(Gxx), d, n, c, _p(W), _p(lam), C.byref(r))
        assert r.value == d
        assert col_rel_err(W, g[f"c{c}/W"]) < 1e-9
    gl = load_golden("losses")
    A = np.ascontiguousarray(gl["inv_sqrtm/A"])
    out = np.zeros_like(A)
    _call(H, "ccz_inv_sqrtm", _p(A), A.shape[0], 1e-5, _p(out))
    np.testing.assert_allclose(out, gl["inv_sqrtm/out_eps1e-5"], atol=1e-9)
    _call(H, "ccz_inv_sqrtm", _p(A), A.shape[0], 0.5, _p(out))
    np.testing.assert_allclose(out, gl["inv_sqrtm/out_eps0.5"], atol=1e-10)
