"""GPU parity: DCCA correlation loss (value + input gradients) against the reference goldens."""

import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _tags(g):
    return sorted({k.rsplit("/", 1)[0] for k in g if k.startswith("cca/")})


def test_cca_loss_value_and_grad_goldens():
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss

    g = load_golden("losses")
    for tag in _tags(g):
        eps = 1e-5 if "unequal" in tag else float(tag.split("eps")[1])
        z1 = torch.tensor(g[tag + "/z1"], device="cuda", requires_grad=True)
        z2 = torch.tensor(g[tag + "/z2"], device="cuda", requires_grad=True)
        loss = CCALoss(eps=eps)([z1, z2])
        assert loss.dim() == 0 and loss.dtype == z1.dtype and loss.is_cuda
        loss.backward()
        f32 = z1.dtype == torch.float32
        ref = float(g[tag + "/loss"])
        assert abs(loss.item() - ref) <= (1e-3 if f32 else 1e-5) * abs(ref), tag
        # the reference's own fp32 autograd-through-eigh gradients are only ~1e-2 accurate at
        # eps=1e-6 (see tests/test_oracle_golden.py); fp32 bar against them is loose, the fp64
        # goldens carry the real pin and the closed-form oracle check below is tight.
        assert rel_err(z1.grad.cpu().numpy(), g[tag + "/g1"]) < (5e-2 if f32 else 1e-5), tag
        assert rel_err(z2.grad.cpu().numpy(), g[tag + "/g2"]) < (5e-2 if f32 else 1e-5), tag


def test_cca_loss_fp32_against_closed_form_oracle():
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss
    from oracle import losses as ol

    torch.manual_seed(0)
    for n, d1, d2 in [(2048, 64, 64), (1000, 96, 40), (8192, 512, 512)]:
        z1 = torch.randn(n, d1)
        z2 = 0.5 * z1[:, :d2] + torch.randn(n, d2) if d2 <= d1 else torch.randn(n, d2)
        a = z1.cuda().requires_grad_(True)
        b = z2.cuda().requires_grad_(True)
        loss = CCALoss(eps=1e-6)([a, b])
        loss.backward()
        l, g1, g2 = ol.cca_loss_closed_form(z1.numpy(), z2.numpy(), 1e-6)
        assert abs(loss.item() - l) <= 1e-3 * abs(l)
        assert rel_err(a.grad.cpu().numpy(), g1) < 1e-3
        assert rel_err(b.grad.cpu().numpy(), g2) < 1e-3


def test_mcca_loss_and_contract():
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss, MCCALoss, _inv_sqrtm

    g = load_golden("losses")
    zs = [torch.tensor(g[f"mcca/z{i}"], device="cuda", requires_grad=True) for i in range(3)]
    loss = MCCALoss(eps=1e-5)(zs)
    loss.backward()
    assert abs(loss.item() - float(g["mcca/loss"])) < 1e-5 * abs(float(g["mcca/loss"]))
    for i in range(3):
        assert rel_err(zs[i].grad.cpu().numpy(), g[f"mcca/g{i}"]) < 1e-5
    with pytest.raises(ValueError, match="exactly 2"):
        CCALoss()([zs[0]])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        CCALoss()([zs[0].detach().cpu(), zs[1].detach().cpu()])
    assert (CCALoss()([zs[0].detach(), zs[1].detach()])).item() <= 0
    A = torch.tensor(g["inv_sqrtm/A"], device="cuda")
    np.testing.assert_allclose(_inv_sqrtm(A, 1e-5).cpu().numpy(), g["inv_sqrtm/out_eps1e-5"], atol=1e-9)
    # repeated eigenvalues: the reference's eigh-autograd gives NaN gradients, the closed form is finite
    q, _ = torch.linalg.qr(torch.randn(64, 8, dtype=torch.float64))
    q = (q - q.mean(0)).cuda().requires_grad_(True)
    r = torch.randn(64, 8, dtype=torch.float64, device="cuda", requires_grad=True)
    CCALoss(eps=1e-4)([q, r]).backward()
    assert torch.isfinite(q.grad).all()


def test_loss_drives_training_step():
    """The objective callable inside an optimisation step (the DCCA training_step contract)."""
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss

    torch.manual_seed(0)
    n = 1024
    zlat = torch.randn(n, 4)
    x1 = (zlat @ torch.randn(4, 20) + 0.5 * torch.randn(n, 20)).cuda()
    x2 = (zlat @ torch.randn(4, 16) + 0.5 * torch.randn(n, 16)).cuda()
    e1, e2 = torch.nn.Linear(20, 4).cuda(), torch.nn.Linear(16, 4).cuda()
    opt = torch.optim.Adam(list(e1.parameters()) + list(e2.parameters()), lr=1e-2)
    obj = CCALoss(eps=1e-4)
    first = None
    for _ in range(30):
        opt.zero_grad()
        loss = obj([e1(x1), e2(x2)])
        loss.backward()
        opt.step()
        first = loss.item() if first is None else first
    assert loss.item() < first - 0.1


@pytest.fixture(scope="module")
def H():
    from cca_zoo_amd import _backend

    return _backend.default_handle(0)


TAG64 = "cca/n256_d32_f64_eps0.0001"


def test_sharded_loss_math_two_shards_one_gpu(H):
    """ccz_cca_loss_moments: moments accumulated over two row shards give the full-batch loss, and Gamma applied to
    each shard gives that shard's rows of the full-batch gradient (goldens of the reference)."""
    import ctypes as C

    from cca_zoo_amd import _backend

    g = load_golden("losses")
    z1, z2, eps = g[TAG64 + "/z1"], g[TAG64 + "/z2"], 1e-4
    n, d1, d2 = z1.shape[0], z1.shape[1], z2.shape[1]
    D = d1 + d2
    zcat = np.ascontiguousarray(np.hstack([z1, z2]))
    cut = n // 3
    mom = H.alloc((D * D + D) * 8)
    H.moments([(zcat[:cut], D, D)], cut, _backend.F64, False, mom.ptr)
    H.moments([(zcat[cut:], D, D)], n - cut, _backend.F64, False, mom.ptr, accumulate=True)
    loss = C.c_double()
    gam, mean = H.alloc(D * D * 8), H.alloc(D * 8)
    H.check(H.lib.ccz_cca_loss_moments(H.raw, C.c_void_p(mom.ptr), n, d1, d2, eps, C.byref(loss),
                                       C.c_void_p(gam.ptr), C.c_void_p(mean.ptr)))
    assert loss.value == pytest.approx(float(g[TAG64 + "/loss"]), rel=1e-9)
    Gamma, mu = H.to_host(gam, (D, D)), H.to_host(mean, (D,))
    np.testing.assert_allclose(mu, zcat.mean(axis=0), rtol=1e-12, atol=1e-14)
    for rows in (slice(0, cut), slice(cut, n)):
        grad = (zcat[rows] - mu) @ Gamma
        np.testing.assert_allclose(grad[:, :d1], g[TAG64 + "/g1"][rows], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(grad[:, d1:], g[TAG64 + "/g2"][rows], rtol=1e-6, atol=1e-9)
    only = C.c_double()
    H.check(H.lib.ccz_cca_loss_moments(H.raw, C.c_void_p(mom.ptr), n, d1, d2, eps, C.byref(only), None, None))
    assert only.value == pytest.approx(loss.value, rel=1e-12)


def test_sharded_loss_module_world_size_one():
    """CCALoss inside row_sharded() (RCCL, world size 1 here): same value and gradients as the fused single-GPU path."""
    import torch
    import torch.distributed as dist

    from cca_zoo_amd import row_sharded
    from cca_zoo_amd.deep import CCALoss

    started = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29653", rank=0, world_size=1)
        started = True
    try:
        g = load_golden("losses")
        for dtype, tol in ((torch.float64, 1e-6), (torch.float32, 1e-3)):
            a = torch.tensor(g[TAG64 + "/z1"], dtype=dtype, device="cuda", requires_grad=True)
            b = torch.tensor(g[TAG64 + "/z2"], dtype=dtype, device="cuda", requires_grad=True)
            with row_sharded():
                loss = CCALoss(eps=1e-4)([a, b])
            loss.backward()
            assert float(loss.detach()) == pytest.approx(float(g[TAG64 + "/loss"]), rel=tol)
            for t, ref in ((a, g[TAG64 + "/g1"]), (b, g[TAG64 + "/g2"])):
                assert np.abs(t.grad.double().cpu().numpy() - ref).max() < tol * np.abs(ref).max()
    finally:
        if started:
            dist.destroy_process_group()
