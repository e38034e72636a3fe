"""GPU parity of PartialCCA / GRCCA against goldens captured from the reference:
one K1 pass over [confounds | views] / the original views, effective moments by device GEMMs, ccz_mcca_solve."""

import numpy as np
import pytest

from conftest import col_rel_err, load_golden

pytestmark = pytest.mark.gpu

PCCA = [("pcca_2v", dict(latent_dimensions=2), 2), ("pcca_3v_ridge", dict(latent_dimensions=2, c=[0.1, 0.3, 0.0]), 3),
        ("pcca_nocenter", dict(latent_dimensions=1, center=False, c=0.2), 2)]
GRCCA = [("grcca_2v", dict(latent_dimensions=2, c=[0.5, 0.8], mu=[0.3, 0.0]), 2),
         ("grcca_3v_mixed", dict(latent_dimensions=2, c=[0.4, 0.0, 0.9], mu=[1.5, 0.2, 0.0]), 3)]


def _align(model, g, tag):
    for i in range(len(model.weights_)):
        s = np.sign(np.sum(model.weights_[i] * g[f"{tag}/w{i}"], axis=0))
        s[s == 0] = 1
        model.weights_[i] = model.weights_[i] * s


@pytest.mark.parametrize("tag,kw,m", PCCA)
def test_partialcca_matches_reference(tag, kw, m):
    from cca_zoo_amd.linear import PartialCCA

    g = load_golden("partial_group")
    views, Z = [g[f"view{i}"] for i in range(m)], g["partials"]
    model = PartialCCA(**kw).fit(views, partials=Z)
    for i in range(m):
        assert model.weights_[i].dtype == np.float64
        assert col_rel_err(model.weights_[i], g[f"{tag}/w{i}"]) < 1e-5
        np.testing.assert_allclose(model.means_[i], g[f"{tag}/mean{i}"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(model.confound_betas_[i], g[f"{tag}/beta{i}"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(model.score(views), g[f"{tag}/score"], rtol=1e-5, atol=1e-7)
    _align(model, g, tag)
    for i, t in enumerate(model.transform(views, partials=Z)):
        np.testing.assert_allclose(t[:6], g[f"{tag}/transform_partials{i}"], rtol=1e-5, atol=1e-6)
    for i, t in enumerate(model.transform(views)):
        np.testing.assert_allclose(t[:6], g[f"{tag}/transform_plain{i}"], rtol=1e-5, atol=1e-6)
    z = PartialCCA(**kw).fit_transform(views, partials=Z)
    assert z[0].shape == (views[0].shape[0], kw["latent_dimensions"])


def test_partialcca_device_tensors_fp32_and_errors():
    import torch

    from cca_zoo_amd.linear import PartialCCA

    g = load_golden("partial_group")
    views, Z = [g[f"view{i}"] for i in range(2)], g["partials"]
    dv = [torch.as_tensor(v, device="cuda") for v in views]
    m64 = PartialCCA(latent_dimensions=2).fit(dv, partials=torch.as_tensor(Z, device="cuda"))
    for i in range(2):
        assert col_rel_err(m64.weights_[i], g[f"pcca_2v/w{i}"]) < 1e-5
    zt = m64.transform(dv, partials=Z)
    assert zt[0].is_cuda and zt[0].shape == (180, 2)
    m32 = PartialCCA(latent_dimensions=2).fit([v.astype(np.float32) for v in views], partials=Z.astype(np.float32))
    for i in range(2):
        assert col_rel_err(m32.weights_[i], g[f"pcca_2v/w{i}"]) < 1e-3
    with pytest.raises(ValueError, match="partials"):
        PartialCCA().fit(views)
    with pytest.raises(ValueError, match="one row per sample"):
        PartialCCA().fit(views, partials=Z[:-1])


@pytest.mark.parametrize("tag,kw,m", GRCCA)
def test_grcca_matches_reference(tag, kw, m):
    from cca_zoo_amd.linear import GRCCA

    g = load_golden("partial_group")
    views = [g[f"view{i}"] for i in range(m)]
    groups = [g[f"groups{i}"] for i in range(m)]
    model = GRCCA(**kw).fit(views, feature_groups=groups)
    for i in range(m):
        assert model.weights_[i].shape == g[f"{tag}/w{i}"].shape
        assert col_rel_err(model.weights_[i], g[f"{tag}/w{i}"]) < 1e-5
    np.testing.assert_allclose(model.score(views), g[f"{tag}/score"], rtol=1e-5, atol=1e-7)
    _align(model, g, tag)
    for i, t in enumerate(model.transform(views)):
        np.testing.assert_allclose(t[:6], g[f"{tag}/transform{i}"], rtol=1e-5, atol=1e-6)


def test_grcca_without_groups_is_mcca_and_warns():
    from cca_zoo_amd.linear import GRCCA, MCCA

    g = load_golden("partial_group")
    views = [g[f"view{i}"] for i in range(2)]
    with pytest.warns(UserWarning, match="No feature_groups"):
        m = GRCCA(latent_dimensions=1, c=0.3).fit(views)
    assert m.weights_[0].shape == (10, 1)
    plain = GRCCA(latent_dimensions=2, c=0.0).fit(views)          # c = 0: no augmentation at all
    ref = MCCA(latent_dimensions=2, c=0.0, pca=False).fit(views)
    for a, b in zip(plain.weights_, ref.weights_):
        assert col_rel_err(a, b) < 1e-8
