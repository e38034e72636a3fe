"""pytest configuration: markers, paths, shared helpers."""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


def sign_align(w, ref):
    """Flip each column of ``w`` to the sign of its inner product with ``ref``."""
    s = np.sign(np.sum(w * ref, axis=0))
    s[s == 0] = 1.0
    return w * s


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def col_rel_err(w, ref):
    """Largest per-column relative error after sign alignment."""
    w = sign_align(np.asarray(w, dtype=np.float64), np.asarray(ref, dtype=np.float64))
    num = np.linalg.norm(w - ref, axis=0)
    den = np.maximum(np.linalg.norm(ref, axis=0), 1e-300)
    return float((num / den).max())


@pytest.fixture(scope="session")
def golden():
    return load_golden


def host_solve_cached(tag, probe, compute):
    """The slow part of the full-size GPU parity tests is not the GPU: it is the ORACLE's dense host solve (LAPACK on
    8192 x 8192 pencils, 40-60 s each) of float64 moments that a deterministic generator reproduces bit for bit.  Its
    result is kept as a fixture (``tests/golden/fullsize/<tag>.npz``, weights stored as float32: 6e-8 relative, the tests
    compare at 1e-5 .. 1e-2) together with a probe of the moments it was computed from; a test first checks that ITS
    moments match the probe (else the fixture is not this data's and the oracle runs live).  ``compute()`` returns a dict
    of arrays.  Fixtures are written by ``tools/gen_golden_fullsize.sh`` (the same tests with CCZ_WRITE_FULLSIZE_GOLDEN
    set to an output directory, on a GPU box)."""
    path = os.path.join(GOLDEN, "fullsize", tag + ".npz")
    probe = np.asarray(probe, dtype=np.float64)
    if os.path.exists(path) and not os.environ.get("CCZ_WRITE_FULLSIZE_GOLDEN"):
        with np.load(path) as f:
            if f["probe"].shape == probe.shape and np.allclose(f["probe"], probe, rtol=1e-11, atol=0.0):
                return {k: (f[k].astype(np.float64) if f[k].dtype == np.float32 else f[k]) for k in f.files if k != "probe"}
    out = compute()
    dest = os.environ.get("CCZ_WRITE_FULLSIZE_GOLDEN")
    if dest:
        os.makedirs(dest, exist_ok=True)
        small = {k: (np.asarray(v, dtype=np.float32) if np.asarray(v).ndim == 2 else np.asarray(v)) for k, v in out.items()}
        np.savez(os.path.join(dest, tag + ".npz"), probe=probe, **small)
    return {k: np.asarray(v, dtype=np.float64) if np.asarray(v).dtype.kind == "f" else np.asarray(v) for k, v in out.items()}


def moments_probe(G, s):
    """A few numbers that pin float64 moments [G | s] (trace, corners, one interior entry, the sum of the column sums)."""
    D = G.shape[0]
    return [float(np.trace(G)), float(G[0, 0]), float(G[D - 1, D - 1]), float(G[D // 2, D // 3]), float(G[1, D - 2]), float(np.sum(s))]
