"""CPU: PartialCCA / GRCCA oracle (reference-structured and second-moment forms) against goldens captured
from the real reference (tests/golden/partial_group.npz)."""

import numpy as np
import pytest

from conftest import col_rel_err, load_golden
from oracle import gram_form as gf
from oracle import partial_group as pg
from oracle import reference_form as rf

PCCA = [("pcca_2v", dict(k=2), 2), ("pcca_3v_ridge", dict(k=2, c=[0.1, 0.3, 0.0]), 3),
        ("pcca_nocenter", dict(k=1, center=False, c=0.2), 2)]
GRCCA = [("grcca_2v", dict(k=2, c=[0.5, 0.8], mu=[0.3, 0.0]), 2),
         ("grcca_3v_mixed", dict(k=2, c=[0.4, 0.0, 0.9], mu=[1.5, 0.2, 0.0]), 3)]


@pytest.mark.parametrize("tag,kw,m", PCCA)
def test_partialcca_oracle(tag, kw, m):
    g = load_golden("partial_group")
    views, Z = [g[f"view{i}"] for i in range(m)], g["partials"]
    kw = dict(kw)
    k = kw.pop("k")
    W, means, betas = pg.partialcca_reference_form(views, Z, k, **kw)
    dims = [v.shape[1] for v in views]
    G, s, n = gf.moments([Z, *views])
    c = kw.get("c", 0.0)
    W2, means2, betas2 = pg.partialcca_from_moments(G, s, n, Z.shape[1], dims, k, c=rf._per_view(c, 0.0, m),
                                                    center=kw.get("center", True))
    for Wx, mx, bx in ((W, means, betas), (W2, means2, betas2)):
        for i in range(m):
            assert col_rel_err(Wx[i], g[f"{tag}/w{i}"]) < 1e-8
            np.testing.assert_allclose(mx[i], g[f"{tag}/mean{i}"], atol=1e-12)
            np.testing.assert_allclose(bx[i], g[f"{tag}/beta{i}"], rtol=1e-8, atol=1e-10)
    # transforms, sign-aligned to the golden weights
    Wa = [w * np.sign(np.sum(w * g[f"{tag}/w{i}"], axis=0)) for i, w in enumerate(W)]
    for i, t in enumerate(pg.partialcca_transform(views, Z, Wa, means, betas)):
        np.testing.assert_allclose(t[:6], g[f"{tag}/transform_partials{i}"], rtol=1e-7, atol=1e-9)
    for i, t in enumerate(rf.project(views, Wa, means)):
        np.testing.assert_allclose(t[:6], g[f"{tag}/transform_plain{i}"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(rf.mean_offdiag_corr(views, W, means), g[f"{tag}/score"], atol=1e-9)


@pytest.mark.parametrize("tag,kw,m", GRCCA)
def test_grcca_oracle(tag, kw, m):
    g = load_golden("partial_group")
    views = [g[f"view{i}"] for i in range(m)]
    groups = [g[f"groups{i}"] for i in range(m)]
    W, means = pg.grcca_reference_form(views, groups, kw["k"], c=kw["c"], mu=kw["mu"])
    G, s, n = gf.moments(views)
    W2, means2 = pg.grcca_from_moments(G, s, n, [v.shape[1] for v in views], groups, kw["k"], c=kw["c"], mu=kw["mu"])
    for Wx in (W, W2):
        for i in range(m):
            assert col_rel_err(Wx[i], g[f"{tag}/w{i}"]) < 1e-8
    np.testing.assert_allclose(rf.mean_offdiag_corr(views, W2, means2), g[f"{tag}/score"], atol=1e-9)
