"""CPU: grid-search oracle (both forms) against the goldens captured from the reference's GridSearchCV,
and the host logic of cca_zoo_amd.model_selection (ranking / cv_results_, generic route) without a GPU."""

import numpy as np
import pytest
from sklearn.base import BaseEstimator

from conftest import load_golden
from oracle import model_selection as oms
from oracle import reference_form as rf

CASES = [
    ("rcca", {"c": [0.0, 0.01, 0.1, 0.5, 0.9], "latent_dimensions": [1, 2]}, 2, 4),
    ("mcca", {"c": [0.0, 0.1, 0.7], "latent_dimensions": [2]}, 3, 3),
    ("gcca", {"c": [0.05, 0.3], "latent_dimensions": [1, 2]}, 3, 3),
]


def _views(g, m):
    return [g[f"view{i}"] for i in range(m)]


@pytest.mark.parametrize("kind,grid,m,cv", CASES)
def test_oracle_grid_search_matches_reference(kind, grid, m, cv):
    g = load_golden("grid_search")
    views = _views(g, m)
    settings, ref_scores = oms.grid_search_reference_form(kind, views, grid, cv)
    names = [repr(sorted(("estimator__" + k, v) for k, v in p.items())) for p in settings]
    assert names == list(g[f"{kind}/params"])                      # ParameterGrid order
    _, mom_scores, _ = oms.grid_search_shared_moments(kind, views, grid, cv)
    for scores in (ref_scores, mom_scores):
        for f in range(cv):
            np.testing.assert_allclose(scores[:, f], g[f"{kind}/split{f}_test_score"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(scores.mean(axis=1), g[f"{kind}/mean_test_score"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(scores.std(axis=1), g[f"{kind}/std_test_score"], rtol=0, atol=1e-9)
        assert int(np.argmax(scores.mean(axis=1))) == int(g[f"{kind}/best_index"])


def test_kfold_bounds_match_sklearn():
    from sklearn.model_selection import KFold

    for n, k in [(240, 4), (241, 4), (10, 3), (7, 7)]:
        ours = oms.kfold_bounds(n, k)
        theirs = [(int(te[0]), int(te[-1]) + 1) for _, te in KFold(k).split(np.zeros((n, 1)))]
        assert ours == theirs


@pytest.mark.parametrize("kind,grid,m,cv", CASES)
def test_cv_results_assembly_matches_reference(kind, grid, m, cv):
    """_finish: mean / std / rank / best_index with scikit-learn's conventions, fed with the golden split scores."""
    from sklearn.model_selection import ParameterGrid

    from cca_zoo_amd.model_selection import GridSearchCV

    g = load_golden("grid_search")
    scores = np.stack([g[f"{kind}/split{f}_test_score"] for f in range(cv)], axis=1)
    gs = GridSearchCV(estimator=None, param_grid=grid, cv=cv)
    gs._finish(list(ParameterGrid(grid)), scores, np.zeros_like(scores), np.zeros_like(scores))
    np.testing.assert_allclose(gs.cv_results_["mean_test_score"], g[f"{kind}/mean_test_score"], atol=1e-15)
    np.testing.assert_allclose(gs.cv_results_["std_test_score"], g[f"{kind}/std_test_score"], atol=1e-15)
    assert list(gs.cv_results_["rank_test_score"]) == list(g[f"{kind}/rank_test_score"])
    assert gs.best_index_ == int(g[f"{kind}/best_index"])
    assert gs.best_score_ == pytest.approx(float(g[f"{kind}/best_score"]), abs=1e-15)
    assert set(gs.cv_results_) >= {"params", "param_c", "param_latent_dimensions", "mean_fit_time", "std_score_time"}
    assert gs.n_splits_ == cv


def test_rank_ties_and_failed_settings():
    from cca_zoo_amd.model_selection import GridSearchCV

    scores = np.array([[0.5, 0.5], [0.7, 0.7], [np.nan, 0.9], [0.7, 0.7]])
    gs = GridSearchCV(None, {"c": [0, 1, 2, 3]}, cv=2)
    gs._finish([{"c": i} for i in range(4)], scores, np.zeros_like(scores), np.zeros_like(scores))
    assert list(gs.cv_results_["rank_test_score"]) == [3, 1, 4, 1]       # rankdata(-mean, "min"), nan last
    assert gs.best_index_ == 1 and gs.best_params_ == {"c": 1}


class _NumpyRCCA(BaseEstimator):
    """A foreign (non-libccz) multiview estimator: must take the generic route."""

    def __init__(self, latent_dimensions=1, c=0.0):
        self.latent_dimensions = latent_dimensions
        self.c = c

    def fit(self, views, y=None):
        self.weights_, self.means_ = rf.rcca_weights(views, self.latent_dimensions, c=self.c)
        return self

    def score(self, views, y=None):
        return rf.mean_offdiag_corr(views, self.weights_, self.means_)

    def transform(self, views):
        return rf.project(views, self.weights_, self.means_)


def test_generic_route_reproduces_reference_scores():
    from cca_zoo_amd.model_selection import GridSearchCV

    g = load_golden("grid_search")
    views = _views(g, 2)
    grid = {"c": [0.0, 0.01, 0.1, 0.5, 0.9], "latent_dimensions": [1, 2]}
    gs = GridSearchCV(_NumpyRCCA(), grid, cv=4).fit(views)
    assert gs.route_ == "generic"
    np.testing.assert_allclose(gs.cv_results_["mean_test_score"], g["rcca/mean_test_score"], atol=1e-9)
    assert gs.best_index_ == int(g["rcca/best_index"])
    assert gs.best_params_ == {"c": 0.9, "latent_dimensions": 1} or gs.best_score_ == pytest.approx(float(g["rcca/best_score"]))
    assert gs.score(views) == pytest.approx(float(g["rcca/score_all"]), abs=1e-9)
    assert gs.transform(views)[0].shape == (240, gs.best_params_["latent_dimensions"])
    with pytest.raises(AttributeError, match="refit=False"):
        GridSearchCV(_NumpyRCCA(), {"c": [0.1]}, cv=2, refit=False).fit(views).transform(views)
