"""GPU parity of GCCALoss and _BatchWhiten against goldens captured from the reference."""

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _zs(g, tag):
    out, i = [], 0
    while f"{tag}/z{i}" in g:
        out.append(g[f"{tag}/z{i}"])
        i += 1
    return out


@pytest.mark.parametrize("tag", ["gcca3", "gcca2_eps", "gcca4"])
@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-5), (torch.float32, 1e-3)])
def test_gcca_loss_matches_reference(tag, dtype, tol):
    from cca_zoo_amd.deep import GCCALoss

    g = load_golden("deep_next")
    zs = [torch.tensor(z, dtype=dtype, device="cuda", requires_grad=True) for z in _zs(g, tag)]
    loss = GCCALoss(eps=float(g[f"{tag}/eps"]))(zs)
    assert loss.dim() == 0 and loss.dtype == dtype and loss.is_cuda
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g[f"{tag}/loss"]), rel=tol)
    for i, z in enumerate(zs):
        ref = g[f"{tag}/g{i}"]
        assert z.grad.dtype == dtype
        assert np.abs(z.grad.double().cpu().numpy() - ref).max() < tol * np.abs(ref).max()


def test_gcca_loss_scales_with_upstream_gradient_and_skips_backward_work():
    from cca_zoo_amd.deep import GCCALoss

    g = load_golden("deep_next")
    zs = [torch.tensor(z, device="cuda", requires_grad=True) for z in _zs(g, "gcca3")]
    (3.0 * GCCALoss(eps=1e-5)(zs)).backward()
    np.testing.assert_allclose(zs[1].grad.cpu().numpy(), 3.0 * g["gcca3/g1"], rtol=1e-5, atol=1e-8)
    with torch.no_grad():
        val = GCCALoss(eps=1e-5)([z.detach() for z in zs])
    assert float(val) == pytest.approx(float(g["gcca3/loss"]), rel=1e-6)
    with pytest.raises(RuntimeError, match="CUDA"):
        GCCALoss()([z.detach().cpu() for z in zs])


def test_batch_whiten_matches_reference_over_three_steps():
    from cca_zoo_amd.deep._dcca_noi import _BatchWhiten

    g = load_golden("deep_next")
    bw = _BatchWhiten(6, momentum=0.2, eps=1e-4).double().cuda()
    bw.train()
    coef = torch.linspace(0.5, 1.5, 6, dtype=torch.float64, device="cuda")
    for step in range(3):
        x = torch.tensor(g[f"bw/x{step}"], device="cuda", requires_grad=True)
        y = bw(x)
        ((y * y) @ coef).sum().backward()
        np.testing.assert_allclose(bw.running_covar.cpu().numpy(), g[f"bw/running{step}"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"bw/y{step}"], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"bw/gx{step}"], rtol=1e-6, atol=1e-8)
    assert int(bw.num_batches_tracked) == int(g["bw/num_batches"])
    bw.eval()
    xe = torch.tensor(g["bw/x2"], device="cuda")
    assert torch.equal(bw(xe), xe)


def test_batch_whiten_fp32_strided_input():
    from cca_zoo_amd.deep._dcca_noi import _BatchWhiten
    from oracle import losses

    rng = np.random.default_rng(3)
    big = rng.standard_normal((500, 40)).astype(np.float32)
    x = torch.tensor(big, device="cuda")[:, 4:20]                  # 16 features, row stride 40
    bw = _BatchWhiten(16, momentum=0.5, eps=1e-3).cuda().train()
    y = bw(x)
    y_ref, running, _ = losses.batch_whiten_step(big[:, 4:20], np.eye(16), 0.5, 1e-3)
    np.testing.assert_allclose(bw.running_covar.cpu().numpy(), running, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref, rtol=2e-3, atol=2e-3)
