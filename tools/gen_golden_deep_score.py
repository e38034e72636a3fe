#!/usr/bin/env python3
"""Golden vector for ``BaseDeep.score``'s arithmetic (cca_zoo/deep/_base.py:159-173: MCCA fit + score on the encoders'
representations), captured from the REAL reference in the build container -> tests/golden/deep_score.npz.
A separate script so that the other goldens are not rewritten; same import shims as tools/gen_golden.py.

    python tools/gen_golden_deep_score.py
"""
import importlib.metadata as md
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference not mounted; goldens can only be regenerated in the build container")
sys.path.insert(0, REF)
_orig_version = md.version
md.version = lambda name: "0.0.0+oracle" if name == "cca_zoo" else _orig_version(name)
_tl = types.ModuleType("tensorly")
_tl.set_backend = lambda *a, **k: None
_dec = types.ModuleType("tensorly.decomposition")
_dec.parafac = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("tensorly stub"))
_tl.decomposition = _dec
sys.modules["tensorly"] = _tl
sys.modules["tensorly.decomposition"] = _dec

from cca_zoo.linear import MCCA  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
rng = np.random.default_rng(2024)
store = {}
for tag, dims, k, n in (("three", (6, 5, 4), 3, 300), ("two", (8, 8), 4, 500)):
    lat = rng.standard_normal((n, k))
    reps = [lat @ rng.standard_normal((k, d)) + 0.7 * rng.standard_normal((n, d)) + 0.2 * i for i, d in enumerate(dims)]
    score = MCCA(latent_dimensions=k).fit(reps).score(reps)          # == BaseDeep.score after transform(loader)
    for i, r in enumerate(reps):
        store[f"{tag}/rep{i}"] = r
    store[f"{tag}/k"] = np.int64(k)
    store[f"{tag}/score"] = np.asarray(score)
np.savez_compressed(os.path.join(OUT, "deep_score.npz"), **store)
print("deep_score", {k: v.shape for k, v in store.items() if k.endswith("score")})
