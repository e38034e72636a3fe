"""scikit-learn's own constructor / get_params / set_params / repr checks over every exported estimator -- the four
checks the reference runs on each of its models (cca_zoo tests/test_sklearn_compat.py:61-75).  They never call fit, so
they run without a GPU; the estimators of this package must pass them exactly as the reference's do."""

import importlib

import pytest
from sklearn.utils.estimator_checks import (
    check_estimator_repr,
    check_get_params_invariance,
    check_no_attributes_set_in_init,
    check_set_params,
)

from cca_zoo_amd._base import BaseModel

_MODULES = ["cca_zoo_amd.linear"]


def _discover():
    found = []
    for name in _MODULES:
        mod = importlib.import_module(name)
        for attr in getattr(mod, "__all__", []):
            obj = getattr(mod, attr)
            if isinstance(obj, type) and issubclass(obj, BaseModel):
                found.append(obj)
    return found


_MODELS = _discover()
_CHECKS = [check_no_attributes_set_in_init, check_get_params_invariance, check_set_params, check_estimator_repr]


@pytest.mark.parametrize("Model", _MODELS, ids=[m.__name__ for m in _MODELS])
@pytest.mark.parametrize("check", _CHECKS, ids=[c.__name__ for c in _CHECKS])
def test_sklearn_estimator_contract(Model, check):
    check(Model.__name__, Model())


def test_every_hot_path_estimator_is_discovered():
    assert {"CCA", "rCCA", "PLS", "MCCA", "GCCA", "GRCCA", "PartialCCA"} <= {m.__name__ for m in _MODELS}
