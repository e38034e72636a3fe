"""CPU tests: the C ABI library loads and exports every declared symbol; host-side logic
(validation, error behaviour, sklearn contract, loud failure without a GPU)."""

import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from cca_zoo_amd.csrc.build import build

    return build()


def test_header_symbols_all_exported(libpath):
    from cca_zoo_amd import _backend

    header = open(os.path.join(ROOT, "include", "ccz.h")).read()
    declared = set(re.findall(r"CCZ_API\s+(?:const\s+char\*|int64_t|int)\s+(ccz_\w+)\s*\(", header))
    assert declared == set(_backend.SIGNATURES), declared ^ set(_backend.SIGNATURES)
    lib = ctypes.CDLL(libpath)
    for name in declared:
        assert hasattr(lib, name), name
    _backend.bind(lib, strict=True)
    assert lib.ccz_version() == 150


def test_no_gpu_fails_loudly(libpath):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cca_zoo_amd.linear import CCA

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        CCA().fit([np.zeros((5, 2)), np.zeros((5, 3))])
    from cca_zoo_amd.deep.objectives import CCALoss

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        CCALoss()([torch.randn(8, 2), torch.randn(8, 2)])


def test_package_never_imports_oracle():
    """The product tree must not reference the test oracle or the host test double."""
    pkg = os.path.join(ROOT, "cca_zoo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
                assert "libccz_hostsim" not in txt or f == "_backend.py", f


def test_validation_and_parameter_constraints():
    from sklearn.utils._param_validation import InvalidParameterError

    from cca_zoo_amd._utils import perview_parameter, validate_views
    from cca_zoo_amd.linear import CCA, GCCA, MCCA, rCCA

    with pytest.raises(ValueError, match="At least 2 views"):
        validate_views([np.zeros((3, 2))])
    with pytest.raises(ValueError, match="same number of samples"):
        validate_views([np.zeros((3, 2)), np.zeros((4, 2))])
    v = validate_views([np.zeros((3, 2), dtype=np.float32), [[1, 2], [3, 4], [5, 6]]])
    assert v[0].dtype == np.float32 and v[1].shape == (3, 2)
    assert perview_parameter("c", None, 0.0, 3) == [0.0] * 3
    assert perview_parameter("c", 0.5, 0.0, 2) == [0.5, 0.5]
    with pytest.raises(ValueError, match="length"):
        perview_parameter("c", [0.1], 0.0, 2)
    X = [np.zeros((6, 2)), np.zeros((6, 3))]
    for bad in (rCCA(latent_dimensions=0), rCCA(center="yes"), rCCA(c=1.5), rCCA(c=-0.1),
                MCCA(pca="no"), MCCA(eps=0.0), GCCA(eps=-1.0), CCA(latent_dimensions=1.5)):
        with pytest.raises(InvalidParameterError):
            bad.fit(X)


def test_sklearn_contract():
    from sklearn.base import clone
    from sklearn.exceptions import NotFittedError

    from cca_zoo_amd.linear import CCA, GCCA, MCCA, PLS, rCCA

    for est in (CCA(2), rCCA(2, c=[0.1, 0.2]), PLS(3, center=False), MCCA(2, c=0.3, pca=False, eps=1e-5),
                GCCA(2, view_weights=[1.0, 2.0])):
        p = est.get_params()
        c = clone(est)
        assert c.get_params() == p and repr(c) == repr(est)
        est.set_params(latent_dimensions=5)
        assert est.latent_dimensions == 5
        with pytest.raises(NotFittedError):
            est.transform([np.zeros((4, 2)), np.zeros((4, 2))])
        with pytest.raises(NotFittedError):
            est.weights
    assert set(CCA().get_params()) == {"latent_dimensions", "center"}
    assert set(rCCA().get_params()) == {"latent_dimensions", "center", "c"}
    assert set(MCCA().get_params()) == {"latent_dimensions", "center", "c", "pca", "eps"}
    assert set(GCCA().get_params()) == {"latent_dimensions", "center", "c", "view_weights", "eps"}


def test_joint_data_matches_reference_stream():
    from conftest import load_golden

    from cca_zoo_amd.datasets import JointData

    g = load_golden("jointdata_seed0")
    jd = JointData(n_views=2, n_samples=200, n_features=[50, 50], latent_dimensions=2, signal_to_noise=2.0,
                   random_state=0)
    a, b = jd.sample(), jd()
    np.testing.assert_array_equal(a[0], g["draw0_v0"])
    np.testing.assert_array_equal(a[1], g["draw0_v1"])
    np.testing.assert_array_equal(b[0], g["draw1_v0"])
    np.testing.assert_array_equal(b[1], g["draw1_v1"])
    with pytest.raises(ValueError, match="n_features"):
        JointData(n_views=2, n_features=[3])


def test_docs_quote_the_real_entry_point_count():
    """DESIGN.md / INTEGRATION.md state how many entry points the ABI has: keep them honest."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = len(re.findall(r"^CCZ_API ", open(os.path.join(root, "include", "ccz.h")).read(), flags=re.M))
    design = open(os.path.join(root, "DESIGN.md")).read()
    integ = open(os.path.join(root, "INTEGRATION.md")).read()
    assert f"({n} `extern \"C\"` entry points" in design
    assert f"argtypes for all {n} symbols" in integ
