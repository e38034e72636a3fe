"""CPU tests for the round-2 additions: certificates, the generator restatement, device-bound handles,
the in-place all-reduce buffer, factor loadings / GCCA loss / GCCA padding through the host double."""

import ctypes as C

import numpy as np
import pytest

from conftest import col_rel_err, load_golden


@pytest.fixture()
def host_handle(monkeypatch):
    from cca_zoo_amd import _backend
    from hostsim_util import hostsim_handle

    h = hostsim_handle()
    monkeypatch.setattr(_backend, "default_handle", lambda device=None: h)
    return h


def _data(seed=0, n=400, dims=(14, 11, 9), k=3):
    from oracle import reference_form as rf

    return rf.joint_data(len(dims), n, k, list(dims), 2.0, seed)


def test_certificates_accept_the_oracle_and_reject_perturbations():
    from oracle import certificates as ct
    from oracle import gram_form as gf

    views = _data(1, 500, (14, 11), 4)
    G, s, n = gf.moments(views)
    dims = [14, 11]
    W, _, sv = gf.rcca_from_moments(G, s, n, dims, 4, c=[0.1, 0.2])
    A, B = ct.rcca_pencil(G, s, n, dims, [0.1, 0.2])
    V = np.vstack(W) / np.sqrt(2.0)
    r = ct.pencil_certificate(A, B, V, sv)
    assert r["residual"] < 1e-12 and r["orthonormality"] < 1e-12 and r["n_above"] == 4
    # dropping the leading pair and keeping 2..5 is still an eigen-system, but not the TOP one
    W5, _, sv5 = gf.rcca_from_moments(G, s, n, dims, 5, c=[0.1, 0.2])
    r = ct.pencil_certificate(A, B, np.vstack([w[:, 1:] for w in W5]) / np.sqrt(2.0), sv5[1:])
    assert r["residual"] < 1e-12 and r["n_above"] == 5 != r["k"]
    # a rotated basis of the same subspace violates the eigen-equation
    Q = np.linalg.qr(np.random.default_rng(0).standard_normal((4, 4)))[0]
    assert ct.pencil_certificate(A, B, V @ Q, sv, inertia=False)["residual"] > 1e-3

    views = _data(2, 600, (14, 11, 9), 3)
    G, s, n = gf.moments(views)
    dims = [14, 11, 9]
    W, _, lam = gf.mcca_from_moments(G, s, n, dims, 3, c=[0.05, 0.1, 0.2])
    A, B = ct.mcca_pencil(G, s, n, dims, [0.05, 0.1, 0.2])
    r = ct.pencil_certificate(A, B, np.vstack(W) / np.sqrt(3.0), lam)
    assert r["residual"] < 1e-12 and r["orthonormality"] < 1e-12 and r["n_above"] == 3
    W, _, lam = gf.mcca_from_moments(G, s, n, dims, 3, c=[0.0, 0.0, 0.0])
    A, B = ct.mcca_pencil(G, s, n, dims, [0.0, 0.0, 0.0])
    r = ct.pencil_certificate(A, B, np.vstack(W) / np.sqrt(3.0), lam)
    assert r["residual"] < 1e-11 and r["n_above"] == 3

    for center, mu in ((True, None), (False, [1.0, 1.0, 2.0])):
        W, _, lam = gf.gcca_from_moments(G, s, n, dims, 3, c=[0.1, 0.1, 0.3], view_weights=mu, center=center)
        A, B, V = ct.gcca_pencil(G, s, n, dims, [0.1, 0.1, 0.3], W, lam, view_weights=mu, center=center)
        r = ct.pencil_certificate(A, B, V, lam)
        assert r["residual"] < 1e-11 and r["orthonormality"] < 1e-11 and r["n_above"] == 3
        Wbad = [w.copy() for w in W]
        Wbad[1][:, 0] *= 1.01
        A, B, V = ct.gcca_pencil(G, s, n, dims, [0.1, 0.1, 0.3], Wbad, lam, view_weights=mu, center=center)
        assert ct.pencil_certificate(A, B, V, lam, inertia=False)["residual"] > 1e-4


def test_generator_restatement_matches_the_shared_hash(host_handle):
    """oracle/rng.py vs csrc/rng_hash.h (compiled into the host double): any row range, odd widths, accumulate."""
    from oracle import rng

    h = host_handle
    for (rows, cols, row0, seed) in ((7, 6, 0, 1), (5, 7, 123456789, 99), (3, 1, 2**33, 2**40 + 3)):
        out = np.full((rows, cols + 2), 7.0)
        stride = cols + (cols & 1)
        h.check(h.lib.ccz_randn_fill(h.raw, 1, C.c_void_p(out.ctypes.data), rows, cols, cols + 2, seed, row0, stride, 1.0, 0))
        ref = rng.randn_block(seed, row0, rows, cols)
        np.testing.assert_allclose(out[:, :cols], ref, rtol=1e-13, atol=1e-15)
        assert np.all(out[:, cols:] == 7.0)
        h.check(h.lib.ccz_randn_fill(h.raw, 1, C.c_void_p(out.ctypes.data), rows, cols, cols + 2, seed + 1, row0, stride, 0.5, 1))
        np.testing.assert_allclose(out[:, :cols], ref + 0.5 * rng.randn_block(seed + 1, row0, rows, cols), rtol=1e-13, atol=1e-15)
    # chunk invariance and fp32 rounding
    a = rng.randn_block(5, 1000, 10, 7)
    b = rng.randn_block(5, 0, 2000, 7)[1000:1010]
    np.testing.assert_array_equal(a, b)
    f = np.zeros((4, 6), dtype=np.float32)
    h.check(h.lib.ccz_randn_fill(h.raw, 0, C.c_void_p(f.ctypes.data), 4, 6, 6, 3, 10, 6, 2.0, 0))
    np.testing.assert_allclose(f, (2.0 * rng.randn_block(3, 10, 4, 6)).astype(np.float32), rtol=1e-6)
    assert h.lib.ccz_randn_fill(h.raw, 1, C.c_void_p(out.ctypes.data), 1, 3, 3, 0, 0, 3, 1.0, 0) != 0     # odd row_stride
    x = rng.randn_block(7, 0, 100000, 32)
    assert abs(x.mean()) < 5e-3 and abs(x.std() - 1.0) < 5e-3
    assert np.abs(np.corrcoef(x[:, :8], rowvar=False) - np.eye(8)).max() < 0.02


def test_joint_data_rows_follow_the_latent_model():
    from cca_zoo_amd.datasets import JointData
    from oracle import rng

    jd = JointData(n_views=2, n_samples=10, latent_dimensions=3, n_features=[5, 4], random_state=0,
                   latent_scales=[2.0, 1.0, 0.5])
    rows = rng.joint_data_rows(jd._weights, jd._snr_per_view, jd.latent_scales, seed=11, row0=100, rows=20000)
    assert rows[0].dtype == np.float32 and rows[0].shape == (20000, 5)
    again = rng.joint_data_rows(jd._weights, jd._snr_per_view, jd.latent_scales, seed=11, row0=100 + 5000, rows=10)
    np.testing.assert_array_equal(rows[0][5000:5010], again[0])
    # covariance of view 0 ~ W diag(s^2) W' + I
    W = jd._weights[0] * np.array([2.0, 1.0, 0.5])
    np.testing.assert_allclose(np.cov(rows[0].astype(np.float64), rowvar=False), W @ W.T + np.eye(5), rtol=0.05, atol=0.1)


def test_handle_for_picks_the_tensors_device(monkeypatch):
    from cca_zoo_amd import _backend

    made = []
    monkeypatch.setattr(_backend, "default_handle", lambda device=None: made.append(device) or ("handle", device))

    class FakeDev:
        def __init__(self, index):
            self.index, self.type = index, "cuda"

    class FakeTensor:
        __module__ = "torch"

        def __init__(self, index):
            self.device, self.is_cuda = FakeDev(index), True

    FakeTensor.__module__ = "torch"
    assert _backend.handle_for([np.zeros(3)]) == ("handle", None)
    assert _backend.handle_for([FakeTensor(1), FakeTensor(1)]) == ("handle", 1)
    assert _backend.handle_for([FakeTensor(3)]) == ("handle", 3)
    with pytest.raises(ValueError, match="same device"):
        _backend.handle_for([FakeTensor(0), FakeTensor(1)])


def test_factor_loadings_and_gcca_padding_on_the_host_double(host_handle):
    from cca_zoo_amd.linear import GCCA, MCCA, rCCA

    views = _data(3, 300, (10, 8), 3)
    m = rCCA(latent_dimensions=3, c=0.1).fit(views)
    fresh = _data(4, 250, (10, 8), 3)
    for given in (views, fresh):
        got = m.get_factor_loadings(given)
        zs = m.transform(given)
        for v, z, l in zip(given, zs, got):
            vc, zc = v - v.mean(0), z - z.mean(0)
            ref = (vc.T @ zc / (len(v) - 1)) / np.outer(vc.std(0, ddof=1), zc.std(0, ddof=1))
            np.testing.assert_allclose(l, ref, atol=1e-10)
    # a constant feature: the reference's max(std, 1e-12) guard -> loading 0, not NaN
    flat = [views[0].copy(), views[1]]
    flat[0][:, 2] = 5.0
    mm = MCCA(latent_dimensions=2, c=0.1).fit(flat)
    l0 = mm.get_factor_loadings(flat)[0]
    assert np.all(np.isfinite(l0)) and np.abs(l0[2]).max() < 1e-6
    # GCCA with more latent dimensions than features: the reference returns min(k, n) columns, the extra ones zero
    g = GCCA(latent_dimensions=30, c=0.1).fit(views)
    assert g.weights_[0].shape == (10, 30) and g.weights_[1].shape == (8, 30)
    assert np.all(g.weights_[0][:, 18:] == 0.0) and np.abs(g.weights_[0][:, :18]).max() > 0
    assert g.score(views).shape == (30,)


def test_gcca_loss_moments_on_the_host_double(host_handle):
    from oracle import gram_form as gf
    from oracle import losses as ol

    h = host_handle
    rng = np.random.default_rng(5)
    zs = [rng.standard_normal((200, d)) + 0.3 * rng.standard_normal((200, 1)) for d in (6, 5, 4)]
    G, s, n = gf.moments(zs)
    D = 15
    mom = np.concatenate([G.ravel(), s])
    dims = (C.c_int64 * 3)(6, 5, 4)
    loss = C.c_double(0.0)
    gam, mean = np.zeros((D, D)), np.zeros(D)
    h.check(h.lib.ccz_gcca_loss_moments(h.raw, C.c_void_p(mom.ctypes.data), n, dims, 3, 1e-4, 6, C.byref(loss),
                                        C.c_void_p(gam.ctypes.data), C.c_void_p(mean.ctypes.data)))
    l_ref, g_ref = ol.gcca_loss_closed_form(zs, 1e-4)
    assert loss.value == pytest.approx(l_ref, rel=1e-9)
    X = np.hstack(zs)
    np.testing.assert_allclose(mean, X.mean(0), atol=1e-12)
    got = (X - mean) @ gam
    np.testing.assert_allclose(got, np.hstack(g_ref), atol=1e-8 * np.abs(np.hstack(g_ref)).max())
    # forward only
    h.check(h.lib.ccz_gcca_loss_moments(h.raw, C.c_void_p(mom.ctypes.data), n, dims, 3, 1e-4, 6, C.byref(loss), None, None))
    assert loss.value == pytest.approx(l_ref, rel=1e-9)


def test_fast_oracle_paths_agree_with_the_dense_ones():
    from oracle import gram_form as gf

    views = _data(7, 900, (40, 30), 5)
    G, s, n = gf.moments(views)
    W, _, sv = gf.rcca_from_moments(G, s, n, [40, 30], 5, c=[0.1, 0.0])
    Wf, _, svf = gf.rcca_from_moments(G, s, n, [40, 30], 5, c=[0.1, 0.0], fast=True)
    np.testing.assert_allclose(svf, sv, rtol=1e-10)
    for a, b in zip(Wf, W):
        assert col_rel_err(a, b) < 1e-8
    views = _data(8, 900, (20, 16, 12), 4)
    G, s, n = gf.moments(views)
    W, _, lam = gf.mcca_from_moments(G, s, n, [20, 16, 12], 4, c=[0.1, 0.1, 0.1])
    Wf, _, lamf = gf.mcca_from_moments(G, s, n, [20, 16, 12], 4, c=[0.1, 0.1, 0.1], fast=True)
    np.testing.assert_allclose(lamf, lam, rtol=1e-10)
    for a, b in zip(Wf, W):
        assert col_rel_err(a, b) < 1e-8


def test_offset_golden_pins_the_oracle_and_the_estimators(host_handle):
    """fp32 views with mean = 100 sigma, golden from the real reference (which centres before any product): the fp64
    second-moment oracle reproduces it, and so do the package's estimators on the host double."""
    from cca_zoo_amd.linear import CCA, rCCA
    from oracle import gram_form as gf

    g = load_golden("offset_two_view_f32")
    train = [g["train0"], g["train1"]]
    assert train[0].dtype == np.float32 and abs(train[0].mean(0) / train[0].std(0)).min() > 50
    G, s, n = gf.moments(train)
    for tag, c in (("rcca_0.1", 0.1), ("cca", 0.0)):
        W, _, _ = gf.rcca_from_moments(G, s, n, [40, 30], 4, c=[c, c])
        for i, w in enumerate(W):
            assert col_rel_err(w, g[f"{tag}/w{i}"].astype(np.float64)) < 1e-4
    for tag, model in (("rcca_0.1", rCCA(latent_dimensions=4, c=0.1)), ("cca", CCA(latent_dimensions=4))):
        model.fit(train)
        for i, w in enumerate(model.weights_):
            assert w.dtype == np.float32 and col_rel_err(w, g[f"{tag}/w{i}"].astype(np.float64)) < 1e-4
        np.testing.assert_allclose(model.score(train), g[f"{tag}/score_train"], atol=1e-4)


def _all_model_classes():
    import cca_zoo_amd.linear as lin
    from cca_zoo_amd._base import BaseModel

    return [getattr(lin, n) for n in getattr(lin, "__all__", dir(lin))
            if isinstance(getattr(lin, n), type) and issubclass(getattr(lin, n), BaseModel) and getattr(lin, n) is not BaseModel]


@pytest.mark.parametrize("check_name", ["check_no_attributes_set_in_init", "check_get_params_invariance", "check_set_params",
                                        "check_estimator_repr"])
def test_sklearn_estimator_contract_sweep(check_name):
    """The reference's sklearn-compat sweep (tests/test_sklearn_compat.py:61-75): the four scikit-learn checks that only
    touch the constructor / get_params / set_params / repr, over every estimator class the package exports."""
    import sklearn.utils.estimator_checks as ec

    classes = _all_model_classes()
    assert {"CCA", "rCCA", "PLS", "MCCA", "GCCA", "PartialCCA", "GRCCA"} <= {c.__name__ for c in classes}
    check = getattr(ec, check_name)
    for cls in classes:
        check(cls.__name__, cls())


def test_grid_search_generic_route_refuses_row_shards(monkeypatch):
    from cca_zoo_amd import _dist
    from cca_zoo_amd.linear import rCCA
    from cca_zoo_amd.model_selection import GridSearchCV

    monkeypatch.setattr(_dist, "is_sharded", lambda: True)
    views = _data(9, 60, (5, 4), 2)
    with pytest.raises(NotImplementedError, match="row_sharded"):
        GridSearchCV(rCCA(latent_dimensions=1), {"c": [0.1]}, cv=3, scoring="r2")._fit_generic(views)


def test_bench_cpu_comparators_run_on_small_shapes():
    """bench.py's CPU legs (the oracle timed as the baseline) on toy sizes: they must run and return finite, positive,
    self-consistent fields -- the full-size runs take tens of seconds and only happen inside bench.py."""
    import bench

    r = bench.cpu_mcca_baseline(n_full=10_000, d=24, m=3, k=4, sample_rows=200)
    assert r["measured_s"] > 0 and r["data_dependent_s"] > 0 and r["eigen_solve_s"] >= 0
    assert abs(r["extrapolated_full_s"] - (r["eigen_solve_s"] + r["data_dependent_s"] * 10_000 / 200)) < 1e-9
    assert abs(r["value"] * r["extrapolated_full_s"] - 1.0) < 1e-12
    b = bench.cpu_baseline(5_000, 16, 3, 128, runs=3)                               # slope from two sample sizes (round 4)
    assert b["kind"] == "port" and b["n_dependent_s"]["rows"] == [128, 256] and b["measured_s"] > 0 and b["runs"] == 3
    assert len(b["n_dependent_s"]["runs_s"][0]) == 3 and len(b["n_dependent_s"]["runs_s"][1]) == 2 and b["per_row_s"] >= 0
    t1, t2 = b["n_dependent_s"]["median_s"]
    assert abs(b["per_row_s"] - max((t2 - t1) / 128, 0.0)) < 1e-12
    assert abs(b["extrapolated_full_s"] - (b["fixed_s"] + t1 + b["per_row_s"] * (5_000 - 128))) < 1e-9
    assert abs(b["value"] * b["extrapolated_full_s"] - 1.0) < 1e-12
    a = bench.cpu_baseline(50_000, 64, 3, runs=1)                                  # default sample: 2 d and 4 d rows
    assert a["n_dependent_s"]["rows"] == [128, 256] and a["runs"] == 1
    g = bench.cpu_gcca_baseline(n_rows=120, dims=(20, 16, 30), k=4)
    assert g["measured_s"] > 0 and g["extrapolated_full_s"] is None
    t, src = bench.gram_traffic("f32", 8192, 1000)
    assert src is not None and "gram_traffic" in src and t > 1000 * 8192 * 4      # more than the algorithmic bytes
