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
