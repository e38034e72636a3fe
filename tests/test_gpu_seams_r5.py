"""Round-5 GPU tests of the dense function seams at the sizes the previous review found untested:
``ccz_whitener`` / ``ccz_inv_sqrtm`` at d = 4096 (the metric's own width: the refresh step of the blocked Jacobi is active),
the Python ``_inv_sqrtm`` on an fp32 512 x 512 CUDA tensor and as a differentiable node, ``_BatchWhiten(512)`` over three
training steps, ``ccz_gevp_topk`` at p = 2048 with a dense SPD ``B``.  Comparators: NumPy / SciPy LAPACK in float64 and the
golden-pinned oracle (oracle/losses.py).  Reference seams: cca_zoo/_utils/_linalg.py:9-73, deep/objectives.py:9-21,
deep/_dcca_noi.py:12-67."""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from cca_zoo_amd import _backend

    return _backend.default_handle(0)


def vp(buf):
    return C.c_void_p(buf.ptr)


def call(H, name, *args):
    H.check(getattr(H.lib, name)(H.raw, *args))


def test_whitener_and_inv_sqrtm_at_the_metric_width(H):
    """d = 4096 against eigh in float64: eigenvalues, the whitening property W' R W = I, A^-1/2 with an active clamp."""
    import torch

    d, n = 4096, 3 * 4096
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randn(n, d, generator=g, device="cuda", dtype=torch.float64) * torch.linspace(1.5, 0.2, d, device="cuda", dtype=torch.float64)
    X -= X.mean(0)
    Gt = X.T @ X                                           # comparator arithmetic (torch), not the product
    G = Gt.cpu().numpy()
    del X, Gt
    Gd = H.to_device(np.ascontiguousarray(G))
    lr = np.linalg.eigvalsh(G / (n - 1))[::-1]
    for c in (0.0, 0.1):
        Wd, ld, r = H.alloc(d * d * 8), H.alloc(d * 8), C.c_int64(0)
        call(H, "ccz_whitener", vp(Gd), d, n, c, vp(Wd), vp(ld), C.byref(r))
        W, lam = H.to_host(Wd, (d, d)), H.to_host(ld, (d,))
        assert r.value == d
        np.testing.assert_allclose(lam, lr, atol=1e-10 * lr[0])
        R = (1 - c) * G / (n - 1) + c * np.eye(d)
        Wt = torch.as_tensor(W, device="cuda")
        dev = (Wt.T @ torch.as_tensor(R, device="cuda") @ Wt - torch.eye(d, device="cuda", dtype=torch.float64)).abs().max().item()
        assert dev < 1e-9, (c, dev)
    A = G / (n - 1)
    Ad, od = H.to_device(np.ascontiguousarray(A)), H.alloc(A.nbytes)
    lam_a, Va = np.linalg.eigh(A)
    for eps in (1e-5, 0.3):                                # 0.3 clamps about a third of the spectrum
        call(H, "ccz_inv_sqrtm", vp(Ad), d, eps, vp(od))
        out = H.to_host(od, (d, d))
        ref = (Va / np.sqrt(np.maximum(lam_a, eps))) @ Va.T
        np.testing.assert_allclose(out, ref, atol=1e-9 * np.abs(ref).max())


def test_python_inv_sqrtm_fp32_at_the_dcca_width():
    """The wrapper the reference's CCALoss.forward would call, on an fp32 512 x 512 batch covariance (configs[3]'s size)."""
    import torch

    from cca_zoo_amd.deep.objectives import _inv_sqrtm

    torch.manual_seed(0)
    z = torch.randn(8192, 512, device="cuda") * torch.linspace(2.0, 0.1, 512, device="cuda") + 0.5
    zc = z - z.mean(0)
    S = (zc.T @ zc) / (z.shape[0] - 1) + 1e-4 * torch.eye(512, device="cuda")
    out = _inv_sqrtm(S, 1e-5)
    assert out.dtype == torch.float32 and out.shape == (512, 512)
    lam, V = np.linalg.eigh(S.double().cpu().numpy())
    ref = (V / np.sqrt(np.maximum(lam, 1e-5))) @ V.T
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=2e-6 * np.abs(ref).max())
    # the defining property in the input's own precision
    I = (out.double() @ S.double() @ out.double()).cpu().numpy()
    assert np.abs(I - np.eye(512)).max() < 1e-4


def test_inv_sqrtm_is_a_differentiable_node():
    """Gradient of a scalar function of A^-1/2 against torch's autograd through the oracle's eigh form (float64), with
    the clamp active on part of the spectrum, and the reference's CCALoss assembled from the seam."""
    import torch

    from cca_zoo_amd.deep.objectives import _inv_sqrtm
    from oracle import losses as ol

    torch.manual_seed(1)
    d = 96
    Q, _ = torch.linalg.qr(torch.randn(d, d, dtype=torch.float64))
    lam = torch.cat([torch.linspace(3.0, 0.05, d - 10, dtype=torch.float64), torch.linspace(5e-4, 1e-6, 10, dtype=torch.float64)])
    A0 = (Q * lam) @ Q.T
    A0 = 0.5 * (A0 + A0.T)
    Rm = torch.randn(d, d, dtype=torch.float64)
    for eps in (1e-8, 1e-3):                               # 1e-3 clamps the ten small eigenvalues
        a_ref = A0.clone().requires_grad_(True)
        (ol.inv_sqrtm_eigh(a_ref, eps) * Rm).sum().backward()
        a_dev = A0.clone().cuda().requires_grad_(True)
        F = _inv_sqrtm(a_dev, eps)
        (F * Rm.cuda()).sum().backward()
        np.testing.assert_allclose(F.detach().cpu().numpy(), ol.inv_sqrtm_eigh(A0, eps).numpy(), atol=1e-9 * float(F.abs().max()))
        g_ref = 0.5 * (a_ref.grad + a_ref.grad.T).numpy()  # eigh's backward reads one triangle; compare symmetric parts
        g_dev = a_dev.grad.cpu().numpy()
        assert np.abs(g_dev - g_ref).max() < 1e-7 * np.abs(g_ref).max(), eps
    # the reference's loss built on the seam: -sum sqrt-eigenvalues ... here simply tr(T'T) with T = S11^-1/2 S12 S22^-1/2
    z1 = torch.randn(400, 24, dtype=torch.float64)
    z2 = 0.6 * z1[:, :16] + torch.randn(400, 16, dtype=torch.float64)

    def loss_of(a, b, isq):
        n = a.shape[0]
        ac, bc = a - a.mean(0), b - b.mean(0)
        S11 = ac.T @ ac / (n - 1) + 1e-4 * torch.eye(a.shape[1], dtype=a.dtype, device=a.device)
        S22 = bc.T @ bc / (n - 1) + 1e-4 * torch.eye(b.shape[1], dtype=b.dtype, device=b.device)
        T = isq(S11) @ (ac.T @ bc / (n - 1)) @ isq(S22)
        return -(T * T).sum()

    a_r, b_r = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    loss_of(a_r, b_r, lambda S: ol.inv_sqrtm_eigh(S, 1e-5)).backward()
    a_d, b_d = z1.clone().cuda().requires_grad_(True), z2.clone().cuda().requires_grad_(True)
    l_dev = loss_of(a_d, b_d, lambda S: _inv_sqrtm(S, 1e-5))
    l_dev.backward()
    np.testing.assert_allclose(a_d.grad.cpu().numpy(), a_r.grad.numpy(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(b_d.grad.cpu().numpy(), b_r.grad.numpy(), rtol=1e-6, atol=1e-9)


def test_batch_whiten_512_over_three_steps():
    """_BatchWhiten at a DCCA width against the golden-pinned oracle step (oracle/losses.py::batch_whiten_step, held to the
    reference's three-step golden in the CPU suite): running matrix, output, input gradient."""
    import torch

    from cca_zoo_amd.deep._dcca_noi import _BatchWhiten
    from oracle import losses

    d, n = 512, 4096
    rng = np.random.default_rng(12)
    bw = _BatchWhiten(d, momentum=0.3, eps=1e-4).double().cuda().train()
    running = np.eye(d)
    coef = np.linspace(0.5, 1.5, d)
    for step in range(3):
        xh = rng.standard_normal((n, d)) * np.linspace(1.5, 0.2, d) + 0.1 * step
        x = torch.tensor(xh, device="cuda", requires_grad=True)
        y = bw(x)
        ((y * y) @ torch.tensor(coef, device="cuda")).sum().backward()
        y_ref, running, w = losses.batch_whiten_step(xh, running, 0.3, 1e-4)
        np.testing.assert_allclose(bw.running_covar.cpu().numpy(), running, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref, rtol=0, atol=1e-8 * np.abs(y_ref).max())
        gx = (2.0 * y_ref * coef) @ w.T                    # no gradient flows through w (as in the reference)
        np.testing.assert_allclose(x.grad.cpu().numpy(), gx, rtol=0, atol=1e-8 * np.abs(gx).max())
    assert int(bw.num_batches_tracked) == 3


def test_gevp_topk_at_2048_with_a_dense_metric(H):
    import scipy.linalg

    p, k = 2048, 32
    rng = np.random.default_rng(2)
    lam = np.concatenate([np.linspace(5.0, 2.0, k), rng.uniform(-3.0, 1.0, p - k)])
    Qm, _ = np.linalg.qr(rng.standard_normal((p, p)))
    A = (Qm * lam) @ Qm.T
    A = 0.5 * (A + A.T)
    Bm = rng.standard_normal((p, p))
    Bm = Bm @ Bm.T / p + np.eye(p)
    wr, Vr = scipy.linalg.eigh(A, Bm, subset_by_index=[p - k, p - 1])
    Ad, Bd, wd, Vd = H.to_device(A), H.to_device(Bm), H.alloc(k * 8), H.alloc(p * k * 8)
    call(H, "ccz_gevp_topk", vp(Ad), vp(Bd), p, k, vp(wd), vp(Vd))
    w, V = H.to_host(wd, (k,)), H.to_host(Vd, (p, k))
    np.testing.assert_allclose(w, wr[::-1], atol=1e-9 * abs(wr).max())
    np.testing.assert_allclose(V.T @ Bm @ V, np.eye(k), atol=1e-9)
    assert np.linalg.norm(A @ V - Bm @ V * w) < 1e-8 * np.linalg.norm(A)
    S = np.sign(np.sum(V * (Bm @ Vr[:, ::-1]), axis=0))
    err = np.linalg.norm(V * S - Vr[:, ::-1], axis=0) / np.linalg.norm(Vr[:, ::-1], axis=0)
    assert err.max() < 1e-6, err.max()


@pytest.mark.parametrize("d1,d2,r0", [(320, 256, 100), (700, 130, 60), (1400, 96, 90)])
def test_rank_detection_between_the_kernels(d1, d2, r0):
    """c = 0 on views that live in an r0-dimensional subspace of their feature space (X_i = Z_i B_i, Z_i of full column rank
    r0 < d_i): the covariances have rank r0 EXACTLY, the Cholesky whitener fails and the eigen-floored fallback must read that
    rank off the block Jacobi's eigenvalues (widths between the one-workgroup kernel and the refresh threshold).
    Canonical correlations are invariant under the maps B_i, so the truth is the full-rank problem (Z_1, Z_2) in float64
    LAPACK; one numerically-zero direction taken for signal would add a spurious correlation and move every value."""
    from cca_zoo_amd.linear import CCA

    n, k = 3000, 5
    rng = np.random.default_rng(d1 + r0)
    Z1 = rng.standard_normal((n, r0))
    Z2 = 0.8 * Z1 @ (rng.standard_normal((r0, r0)) / np.sqrt(r0)) + rng.standard_normal((n, r0))
    X1 = Z1 @ rng.standard_normal((r0, d1))
    X2 = Z2 @ rng.standard_normal((r0, d2))
    m = CCA(latent_dimensions=k).fit([X1, X2])
    q1, _ = np.linalg.qr(Z1 - Z1.mean(0))
    q2, _ = np.linalg.qr(Z2 - Z2.mean(0))
    truth = np.linalg.svd(q1.T @ q2, compute_uv=False)[:k]
    np.testing.assert_allclose(np.asarray(m.singular_values_)[:k], truth, rtol=0, atol=1e-7)
    zs = m.transform([X1, X2])
    np.testing.assert_allclose(np.var(zs[0], axis=0, ddof=1), 1.0, atol=1e-6)
    for a in range(k):
        assert abs(np.corrcoef(zs[0][:, a], zs[1][:, a])[0, 1] - truth[a]) < 1e-6
