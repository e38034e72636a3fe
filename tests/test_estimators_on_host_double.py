"""CPU: the package's HOST LOGIC end to end on the host test double of libccz.

`tests/hostsim` compiles the product driver source (csrc/solve.cpp) against host loops and adds plain-loop doubles
for the HIP-side entry points (K1, memory, transform).  With `_backend.default_handle` pointed at it, the real
estimator / grid-search / partial / group code of `cca_zoo_amd` runs here without a GPU and is held to the goldens
captured from the reference -- the same assertions as the `-m gpu` suites, minus the kernels.  The package never
loads this double itself (`test_abi_and_host_logic.py` checks that a missing GPU library raises).
"""

import numpy as np
import pytest

from conftest import col_rel_err, load_golden
from hostsim_util import hostsim_handle


@pytest.fixture()
def host_double(monkeypatch):
    from cca_zoo_amd import _backend

    h = hostsim_handle()
    monkeypatch.setattr(_backend, "default_handle", lambda device=None: h)
    return h


def _views(g, prefix="view"):
    out, i = [], 0
    while f"{prefix}{i}" in g:
        out.append(g[f"{prefix}{i}"])
        i += 1
    return out


def _specs():
    from cca_zoo_amd.linear import CCA, GCCA, MCCA, PLS, rCCA

    return {
        "cca": lambda: CCA(latent_dimensions=2),
        "rcca_0.1": lambda: rCCA(latent_dimensions=2, c=0.1),
        "rcca_0.1_0.3": lambda: rCCA(latent_dimensions=2, c=[0.1, 0.3]),
        "pls": lambda: PLS(latent_dimensions=2),
        "rcca_0.1_nocenter": lambda: rCCA(latent_dimensions=2, c=0.1, center=False),
        "mcca_c0_pca": lambda: MCCA(latent_dimensions=2, c=0.0, pca=True),
        "mcca_c0.1_nopca": lambda: MCCA(latent_dimensions=2, c=0.1, pca=False),
        "mcca_c0.1_nocenter": lambda: MCCA(latent_dimensions=2, c=0.1, center=False),
        "gcca_c0": lambda: GCCA(latent_dimensions=2, c=0.0),
        "gcca_c0.1": lambda: GCCA(latent_dimensions=2, c=0.1),
        "gcca_c0.1_nocenter": lambda: GCCA(latent_dimensions=2, c=0.1, center=False),
    }


@pytest.mark.parametrize("tag", ["cca", "rcca_0.1", "rcca_0.1_0.3", "pls", "rcca_0.1_nocenter", "mcca_c0_pca",
                                 "mcca_c0.1_nopca", "mcca_c0.1_nocenter", "gcca_c0", "gcca_c0.1", "gcca_c0.1_nocenter"])
def test_c1_configuration_on_host_double(host_double, tag):
    """BASELINE configs[0] and its siblings: the whole estimator surface against the reference's outputs."""
    g = load_golden("c1_two_view_f64")
    train, fresh = _views(g, "train"), _views(g, "fresh")
    model = _specs()[tag]().fit(train)
    assert model.n_views_ == 2 and model.n_features_in_ == [50, 50] and model.n_samples_ == 200
    for i, w in enumerate(model.weights_):
        assert w.dtype == g[f"{tag}/w{i}"].dtype
        assert col_rel_err(w, g[f"{tag}/w{i}"]) < 1e-6
        np.testing.assert_allclose(model.means_[i], g[f"{tag}/mean{i}"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(model.score(train), g[f"{tag}/score_train"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(model.score(fresh), g[f"{tag}/score_fresh"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(model.pairwise_correlations(train), g[f"{tag}/pairwise_train"], rtol=1e-6, atol=1e-8)
    for i in range(2):
        s = np.sign(np.sum(model.weights_[i] * g[f"{tag}/w{i}"], axis=0))
        model.weights_[i] = model.weights_[i] * s
    for i, t in enumerate(model.transform(train)):
        np.testing.assert_allclose(t[:5], g[f"{tag}/transform{i}"], rtol=1e-6, atol=1e-8)
    for i, l in enumerate(model.get_factor_loadings(train)):
        np.testing.assert_allclose(l, g[f"{tag}/loadings{i}"], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("kind,grid,m,cv", [
    ("rcca", {"c": [0.0, 0.01, 0.1, 0.5, 0.9], "latent_dimensions": [1, 2]}, 2, 4),
    ("mcca", {"c": [0.0, 0.1, 0.7], "latent_dimensions": [2]}, 3, 3),
    ("gcca", {"c": [0.05, 0.3], "latent_dimensions": [1, 2]}, 3, 3),
])
def test_grid_search_shared_moments_on_host_double(host_double, kind, grid, m, cv):
    from cca_zoo_amd.linear import GCCA, MCCA, rCCA
    from cca_zoo_amd.model_selection import GridSearchCV

    g = load_golden("grid_search")
    views = [g[f"view{i}"] for i in range(m)]
    est = {"rcca": rCCA, "mcca": MCCA, "gcca": GCCA}[kind]()
    gs = GridSearchCV(est, grid, cv=cv).fit(views)
    assert gs.route_ == "shared-moments"
    for f in range(cv):
        np.testing.assert_allclose(gs.cv_results_[f"split{f}_test_score"], g[f"{kind}/split{f}_test_score"], rtol=1e-7, atol=1e-9)
    assert list(gs.cv_results_["rank_test_score"]) == list(g[f"{kind}/rank_test_score"])
    assert gs.best_index_ == int(g[f"{kind}/best_index"])
    for i, w in enumerate(gs.best_estimator_.weights_):
        assert col_rel_err(w, g[f"{kind}/best_w{i}"]) < 1e-7
    assert gs.score(views) == pytest.approx(float(g[f"{kind}/score_all"]), rel=1e-7)


@pytest.mark.parametrize("tag,kw,m", [("pcca_2v", dict(latent_dimensions=2), 2),
                                      ("pcca_3v_ridge", dict(latent_dimensions=2, c=[0.1, 0.3, 0.0]), 3),
                                      ("pcca_nocenter", dict(latent_dimensions=1, center=False, c=0.2), 2)])
def test_partialcca_on_host_double(host_double, tag, kw, m):
    from cca_zoo_amd.linear import PartialCCA

    g = load_golden("partial_group")
    views, Z = [g[f"view{i}"] for i in range(m)], g["partials"]
    model = PartialCCA(**kw).fit(views, partials=Z)
    for i in range(m):
        assert col_rel_err(model.weights_[i], g[f"{tag}/w{i}"]) < 1e-7
        np.testing.assert_allclose(model.confound_betas_[i], g[f"{tag}/beta{i}"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(model.means_[i], g[f"{tag}/mean{i}"], atol=1e-12)
    np.testing.assert_allclose(model.score(views), g[f"{tag}/score"], rtol=1e-7, atol=1e-9)
    for i in range(m):                                                   # sign-align, then compare projections
        s = np.sign(np.sum(model.weights_[i] * g[f"{tag}/w{i}"], axis=0))
        model.weights_[i] = model.weights_[i] * s
    for i, t in enumerate(model.transform(views, partials=Z)):
        np.testing.assert_allclose(t[:6], g[f"{tag}/transform_partials{i}"], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("tag,kw,m", [("grcca_2v", dict(latent_dimensions=2, c=[0.5, 0.8], mu=[0.3, 0.0]), 2),
                                      ("grcca_3v_mixed", dict(latent_dimensions=2, c=[0.4, 0.0, 0.9], mu=[1.5, 0.2, 0.0]), 3)])
def test_grcca_on_host_double(host_double, tag, kw, m):
    from cca_zoo_amd.linear import GRCCA

    g = load_golden("partial_group")
    views = [g[f"view{i}"] for i in range(m)]
    model = GRCCA(**kw).fit(views, feature_groups=[g[f"groups{i}"] for i in range(m)])
    for i in range(m):
        assert col_rel_err(model.weights_[i], g[f"{tag}/w{i}"]) < 1e-7
    np.testing.assert_allclose(model.score(views), g[f"{tag}/score"], rtol=1e-7, atol=1e-9)


def test_nan_input_is_reported_from_the_column_sums(host_double):
    from cca_zoo_amd.linear import rCCA

    x = np.random.default_rng(0).standard_normal((40, 3))
    y = x + 0.1
    x[7, 1] = np.inf
    with pytest.raises(ValueError, match="NaN or infinity"):
        rCCA().fit([x, y])


def test_grid_search_and_partial_edge_cases_on_host_double(host_double):
    from sklearn.model_selection import KFold

    from cca_zoo_amd.linear import CCA, GRCCA, MCCA, PartialCCA, rCCA
    from cca_zoo_amd.model_selection import GridSearchCV

    rng = np.random.default_rng(0)
    X = rng.standard_normal((60, 5))
    Y = X @ rng.standard_normal((5, 4)) + 0.5 * rng.standard_normal((60, 4))
    z = rng.standard_normal(60)                                    # 1-D confound
    m = PartialCCA(latent_dimensions=2).fit([X, Y], partials=z)
    assert [b.shape for b in m.confound_betas_] == [(1, 5), (1, 4)]
    assert m.transform([X, Y], partials=z)[0].shape == (60, 2)
    m32 = PartialCCA().fit([X.astype(np.float32), Y.astype(np.float32)], partials=z)
    assert m32.weights_[0].dtype == np.float64                     # MCCA family promotes (np.cov)
    labels = [np.array(list("aabbc")), np.array([1, 1, 2, 2])]     # any hashable group labels
    assert GRCCA(latent_dimensions=1, c=0.5, mu=0.1).fit([X, Y], feature_groups=labels).weights_[1].shape == (4, 1)

    gs = GridSearchCV(rCCA(), {"c": [0.1, 0.5]}, cv=3, refit=False).fit([X, Y])
    assert not hasattr(gs, "best_estimator_") and gs.best_params_["c"] in (0.1, 0.5)
    with pytest.raises(AttributeError):
        gs.score([X, Y])
    gs = GridSearchCV(rCCA(), [{"c": [0.1]}, {"latent_dimensions": [1, 2]}], cv=2).fit([X, Y])
    assert len(gs.cv_results_["params"]) == 3 and gs.cv_results_["param_c"].mask.tolist() == [False, True, True]
    gs = GridSearchCV(MCCA(), {"c": [0.2]}, cv=KFold(4)).fit([X, Y])
    assert gs.route_ == "shared-moments" and gs.n_splits_ == 4
    gs = GridSearchCV(CCA(), {"latent_dimensions": [1, 9]}, cv=2).fit([X, Y])          # k clamps to the view width
    assert gs.best_estimator_.weights_[0].shape[1] <= 5
    with pytest.warns(RuntimeWarning, match="fit failed"), pytest.raises(ValueError, match="All the 2 fits failed"):
        GridSearchCV(rCCA(), {"c": [2.0]}, cv=2).fit([X, Y])


@pytest.mark.parametrize("tag", ["three", "two"])
def test_deep_score_of_representations_matches_the_reference(host_double, tag):
    """``BaseDeep.score`` (cca_zoo/deep/_base.py:159-173) = MCCA fit + score on the encoders' representations: the golden
    holds representations and the reference's score (tools/gen_golden_deep_score.py)."""
    from cca_zoo_amd.deep import score_representations

    g = load_golden("deep_score")
    reps, i = [], 0
    while f"{tag}/rep{i}" in g:
        reps.append(g[f"{tag}/rep{i}"])
        i += 1
    got = score_representations(reps, int(g[f"{tag}/k"]))
    assert got.shape == g[f"{tag}/score"].shape
    np.testing.assert_allclose(got, g[f"{tag}/score"], rtol=1e-8, atol=1e-10)
    with pytest.raises(ValueError, match="two views"):
        score_representations(reps[:1], 2)
