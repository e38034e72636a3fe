"""Oracle for SURVEY.md 8 row f2: grid search with cross-validation (cca_zoo/model_selection/_search.py:146-306).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Two layers, as for the estimators:

* ``grid_search_reference_form``: what the reference computes -- scikit-learn's unshuffled K-fold
  (``_search.py:229-238`` passes ``cv`` straight to ``sklearn.model_selection.GridSearchCV``), one fit
  from the TRAINING ROWS per (setting, fold) with ``oracle.reference_form`` and the wrapper's score
  (``_search.py:70-83``: mean over latent dimensions of ``estimator.score``) on the held-out rows;
* ``grid_search_shared_moments``: what ``cca_zoo_amd.model_selection.GridSearchCV`` computes -- one set of
  moments per fold, training moments by subtraction, ``oracle.gram_form`` solves, scores from the held-out
  fold's moments.

Both are pinned by ``tests/golden/grid_search.npz`` (captured from the real reference).
"""

from __future__ import annotations

import itertools

import numpy as np

from oracle import gram_form as gf
from oracle import reference_form as rf


def kfold_bounds(n, n_splits):
    """Test-fold row ranges of ``sklearn.model_selection.KFold(n_splits)`` (no shuffle): the first
    ``n % n_splits`` folds get one extra row."""
    sizes = np.full(n_splits, n // n_splits)
    sizes[: n % n_splits] += 1
    edges = np.concatenate([[0], np.cumsum(sizes)])
    return [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:])]


def parameter_grid(grid):
    """Settings in scikit-learn's ``ParameterGrid`` order: keys sorted, last key varies fastest."""
    keys = sorted(grid)
    return [dict(zip(keys, vals)) for vals in itertools.product(*(grid[k] for k in keys))]


_FITS = {
    "rcca": lambda v, p: rf.rcca_weights(v, p.get("latent_dimensions", 1), c=p.get("c", 0.0)),
    "mcca": lambda v, p: rf.mcca_weights(v, p.get("latent_dimensions", 1), c=p.get("c", 0.0)),
    "gcca": lambda v, p: rf.gcca_weights(v, p.get("latent_dimensions", 1), c=p.get("c", 0.0)),
}


def grid_search_reference_form(kind, views, grid, cv):
    views = [np.asarray(v, dtype=np.float64) for v in views]
    n = views[0].shape[0]
    settings = parameter_grid(grid)
    scores = np.zeros((len(settings), cv))
    for f, (a, b) in enumerate(kfold_bounds(n, cv)):
        train = [np.vstack([v[:a], v[b:]]) for v in views]
        test = [v[a:b] for v in views]
        for i, p in enumerate(settings):
            W, means = _FITS[kind](train, p)
            scores[i, f] = rf.mean_offdiag_corr(test, W, means).mean()
    return settings, scores


def _score_from_moments(G, s, n, dims, W):
    m, k = len(dims), W[0].shape[1]
    D = sum(dims)
    Wbig = np.zeros((D, m * k))
    o = 0
    for i, (d, w) in enumerate(zip(dims, W)):
        Wbig[o:o + d, i * k:(i + 1) * k] = w
        o += d
    sw = s @ Wbig
    S = Wbig.T @ G @ Wbig - np.outer(sw, sw) / n
    nrm = np.sqrt(np.maximum(np.diag(S), 0.0))
    nrm = np.where(nrm > 1e-12, nrm, 1.0)
    R = S / np.outer(nrm, nrm)
    per_dim = sum(np.diag(R[i * k:(i + 1) * k, j * k:(j + 1) * k]) for i in range(m) for j in range(m) if i != j)
    return float(np.mean(per_dim / (m * (m - 1))))


def grid_search_shared_moments(kind, views, grid, cv):
    views = [np.asarray(v, dtype=np.float64) for v in views]
    n = views[0].shape[0]
    dims = [v.shape[1] for v in views]
    settings = parameter_grid(grid)
    folds = [(b - a,) + gf.moments([v[a:b] for v in views])[:2] for a, b in kfold_bounds(n, cv)]
    G_all = sum(f[1] for f in folds)
    s_all = sum(f[2] for f in folds)
    scores = np.zeros((len(settings), cv))
    for f, (n_f, G_f, s_f) in enumerate(folds):
        G_tr, s_tr, n_tr = G_all - G_f, s_all - s_f, n - n_f
        for i, p in enumerate(settings):
            k = p.get("latent_dimensions", 1)
            c = [p.get("c", 0.0)] * len(dims)
            if kind == "rcca":
                W = gf.rcca_from_moments(G_tr, s_tr, n_tr, dims, k, c=tuple(c))[0]
            elif kind == "mcca":
                W = gf.mcca_from_moments(G_tr, s_tr, n_tr, dims, k, c=c)[0]
            else:
                W = gf.gcca_from_moments(G_tr, s_tr, n_tr, dims, k, c=c)[0]
            scores[i, f] = _score_from_moments(G_f, s_f, n_f, dims, W)
    return settings, scores, (G_all, s_all)
