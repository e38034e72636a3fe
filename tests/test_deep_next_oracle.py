"""CPU: oracle for GCCALoss / _BatchWhiten against goldens captured from the reference."""

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import losses


def _zs(g, tag):
    out, i = [], 0
    while f"{tag}/z{i}" in g:
        out.append(g[f"{tag}/z{i}"])
        i += 1
    return out


@pytest.mark.parametrize("tag", ["gcca3", "gcca2_eps", "gcca4"])
def test_gcca_loss_oracle(tag):
    g = load_golden("deep_next")
    zs, eps = _zs(g, tag), float(g[f"{tag}/eps"])
    ts = [torch.tensor(z, requires_grad=True) for z in zs]
    loss = losses.gcca_loss_autograd(ts, eps)
    loss.backward()
    assert float(loss) == pytest.approx(float(g[f"{tag}/loss"]), rel=1e-10)
    for i, t in enumerate(ts):
        np.testing.assert_allclose(t.grad.numpy(), g[f"{tag}/g{i}"], rtol=1e-7, atol=1e-9)
    val, grads = losses.gcca_loss_closed_form(zs, eps)
    assert val == pytest.approx(float(g[f"{tag}/loss"]), rel=1e-9)
    for i, gr in enumerate(grads):
        scale = np.abs(g[f"{tag}/g{i}"]).max()
        assert np.abs(gr - g[f"{tag}/g{i}"]).max() < 1e-7 * scale


def test_batch_whiten_oracle():
    g = load_golden("deep_next")
    running = np.eye(6)
    coef = np.linspace(0.5, 1.5, 6)
    for step in range(3):
        x = g[f"bw/x{step}"]
        y, running, w = losses.batch_whiten_step(x, running, 0.2, 1e-4)
        np.testing.assert_allclose(running, g[f"bw/running{step}"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(y, g[f"bw/y{step}"], rtol=1e-9, atol=1e-11)
        gx = (2.0 * y * coef) @ w.T                      # d/dx of sum((x w)^2 coef), w held constant
        np.testing.assert_allclose(gx, g[f"bw/gx{step}"], rtol=1e-9, atol=1e-11)
    assert float(g["bw/eval_identity"]) == 0.0 and int(g["bw/num_batches"]) == 3
