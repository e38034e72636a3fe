"""Round-5 GPU tests: the persistent Cholesky chain (csrc/cholinv.hip: k_cholinv_chain, 16-column panels), the loss
fast path (K1 partial sums -> k_loss_prep_partials -> chain -> product stages with the loss riding on the third ->
k_loss_tail) and the two-phase loss ABI (ccz_pair_loss_forward / _backward: the upstream gradient applied inside the
sample-side product).  Reference seams: cca_zoo/deep/objectives.py:61-102 (CCALoss.forward + autograd), :138-153."""

import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from cca_zoo_amd import _backend

    return _backend.default_handle()


def _spd(rng, d, cond=1e3):
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    w = np.exp(np.linspace(0.0, np.log(cond), d))
    return (q * w) @ q.T


def _cholinv(H, mats, want_x=True):
    bufs, pa, pl, px = [], [], [], []
    for A in mats:
        d = A.shape[0]
        a, l, x = H.to_device(A), H.to_device(np.full((d, d), np.nan)), H.to_device(np.zeros((d, d)))
        bufs.append((a, l, x))
        pa.append(a.ptr), pl.append(l.ptr), px.append(x.ptr)
    n = len(mats)
    arr = lambda p: (C.c_void_p * n)(*p)
    dd = (C.c_int64 * n)(*[m.shape[0] for m in mats])
    H.check(H.lib.ccz_cholinv(H.raw, n, arr(pa), dd, arr(pl), arr(px) if want_x else None))
    return [(H.to_host(l, A.shape), H.to_host(x, A.shape)) for (a, l, x), A in zip(bufs, mats)]


# ---------------------------------------------------------------------------------------------
# the chain kernel
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [64, 65, 127, 128, 192, 448, 512, 513, 1024, 1500, 2048])
def test_chain_single_matrix(H, d):
    rng = np.random.default_rng(1000 + d)
    A = _spd(rng, d, 1e5)
    (L, X), = _cholinv(H, [A])
    Lr = np.linalg.cholesky(A)
    assert np.abs(np.tril(L) - Lr).max() < 2e-11 * np.abs(Lr).max()
    assert np.all(np.isnan(L[np.triu_indices(d, 1)]))            # strictly-upper part untouched
    Xr = np.linalg.inv(Lr)
    assert np.abs(np.tril(X) - Xr).max() < 1e-9 * np.abs(Xr).max()
    for bi in range(0, d, 64):                                    # blocks above the block diagonal: left alone
        assert np.all(X[bi:bi + 64, bi + 64:] == 0.0)


def test_chain_repeated_launches_leave_the_sync_block_clean(H):
    """The last workgroup of a launch clears the progress counters: twenty launches in a row (different shapes in
    between) must all be right."""
    rng = np.random.default_rng(5)
    for it in range(20):
        ds = [512, 512] if it % 3 else [200, 64, 700]
        mats = [_spd(rng, d, 1e4) for d in ds]
        for (L, X), A in zip(_cholinv(H, mats), mats):
            Lr = np.linalg.cholesky(A)
            assert np.abs(np.tril(L) - Lr).max() < 1e-10 * np.abs(Lr).max()
            assert np.abs(np.tril(X) @ Lr - np.eye(A.shape[0])).max() < 1e-8


def test_chain_eight_matrices_and_failure(H):
    rng = np.random.default_rng(9)
    mats = [_spd(rng, d, 1e4) for d in (512, 512, 512, 512, 333, 512, 129, 512)]
    for (L, X), A in zip(_cholinv(H, mats), mats):
        Lr = np.linalg.cholesky(A)
        assert np.abs(np.tril(L) - Lr).max() < 1e-10 * np.abs(Lr).max()
        assert np.abs(np.tril(X) @ Lr - np.eye(A.shape[0])).max() < 1e-8
    (L, X), = _cholinv(H, [mats[0]], want_x=False)
    assert np.abs(np.tril(L) - np.linalg.cholesky(mats[0])).max() < 1e-10 and np.all(X == 0.0)
    bad = mats[4].copy()
    bad[150, 150] = -1.0
    with pytest.raises(np.linalg.LinAlgError, match="not positive definite"):
        _cholinv(H, [mats[1], bad])
    # ... and the launch after a failed one is clean again
    (L, X), = _cholinv(H, [mats[6]])
    assert np.abs(np.tril(L) - np.linalg.cholesky(mats[6])).max() < 1e-10


_CHILD = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import torch
from cca_zoo_amd import _backend
from cca_zoo_amd.deep.objectives import CCALoss
from oracle import losses as ol
H = _backend.default_handle()
rng = np.random.default_rng(3)
def spd(d):
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    return (q * np.exp(np.linspace(0.0, np.log(1e4), d))) @ q.T
for ds in ([512, 512], [1000], [64, 200, 448, 512, 130, 512, 70, 512]):
    mats = [spd(d) for d in ds]
    bufs = [(H.to_device(A), H.to_device(np.zeros(A.shape)), H.to_device(np.zeros(A.shape))) for A in mats]
    n = len(mats)
    arr = lambda i: (C.c_void_p * n)(*[b[i].ptr for b in bufs])
    H.check(H.lib.ccz_cholinv(H.raw, n, arr(0), (C.c_int64 * n)(*ds), arr(1), arr(2)))
    for (a, l, x), A in zip(bufs, mats):
        Lr = np.linalg.cholesky(A)
        L, X = H.to_host(l, A.shape), H.to_host(x, A.shape)
        assert np.abs(np.tril(L) - Lr).max() < 1e-10 * np.abs(Lr).max(), ds
        assert np.abs(np.tril(X) @ Lr - np.eye(A.shape[0])).max() < 1e-8, ds
g = torch.Generator().manual_seed(11)
for n_, d1, d2 in ((8192, 512, 512), (3000, 256, 768), (1000, 100, 37)):
    z1 = torch.randn(n_, d1, generator=g, dtype=torch.float64)
    z2 = 0.7 * z1 @ (torch.randn(d1, d2, generator=g, dtype=torch.float64) / d1 ** 0.5) + torch.randn(n_, d2, generator=g, dtype=torch.float64) + 0.5
    a = z1.float().cuda().requires_grad_(True); b = z2.float().cuda().requires_grad_(True)
    loss = CCALoss(eps=1e-5)([a, b]); (2.5 * loss).backward()
    l, g1, g2 = ol.cca_loss_closed_form(a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy(), 1e-5)
    assert abs(loss.item() - l) <= 1e-3 * abs(l), (loss.item(), l)
    e1 = np.linalg.norm(a.grad.cpu().numpy() - 2.5 * g1) / np.linalg.norm(2.5 * g1)
    e2 = np.linalg.norm(b.grad.cpu().numpy() - 2.5 * g2) / np.linalg.norm(2.5 * g2)
    assert e1 < 1e-2 and e2 < 1e-2, (e1, e2)
print("child ok")
"""


@pytest.mark.parametrize("env", [
    {"CCZ_CHOLINV_CHAIN": "0"},                           # launch-per-link form, 16-column panels
    {"CCZ_CHOLINV_CHAIN": "0", "CCZ_CHOLINV_MFMA": "1"},  # ... with the 4-column panels of rounds 2-4
    {"CCZ_CHOLINV_MFMA": "1"},                            # chain kernel on the 4-column panels
    {"CCZ_CHAIN_WGS": "2"},                               # chain workgroups + ONE helper: progress must not need co-residency
    {"CCZ_CHAIN_WGS": "24"},
    {"CCZ_LOSS_FAST": "0"},                               # the general route of the loss (moments + k_loss_prep)
    {"CCZ_LOSS_SPLITK": "1"},
])
def test_switches_in_a_child_process(env):
    # (one child at a time: seven processes sharing the GPU with their persistent chain kernels did not finish in 10 minutes)
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, os.path.join(ROOT, "tests"))], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---------------------------------------------------------------------------------------------
# two-phase loss
# ---------------------------------------------------------------------------------------------
def _views(ts):
    from cca_zoo_amd import _backend

    v = (_backend.View * len(ts))()
    for i, t in enumerate(ts):
        v[i].data, v[i].cols, v[i].ld = t.data_ptr(), int(t.shape[1]), int(t.stride(0))
    return v


@pytest.mark.parametrize("n,dims,dtype,tol", [
    (4096, (512, 512), "f32", 1e-3),          # both phases on the fast path: partial sums in, two-source FIFO product out
    (2000, (256, 512), "f32", 1e-3),          # unequal aligned widths (different leading dimensions of the two sources)
    (1500, (100, 37), "f32", 1e-3),           # ragged: partial-sum preparation, general backward
    (1200, (96, 40), "f64", 1e-9),
    (1500, (64, 40, 72), "f32", 1e-3),        # three views
])
def test_two_phase_abi_against_the_closed_form(H, n, dims, dtype, tol):
    import torch

    from cca_zoo_amd import _backend
    from oracle import losses as ol

    torch.manual_seed(n + sum(dims))
    tdt = torch.float64 if dtype == "f64" else torch.float32
    code = _backend.F64 if dtype == "f64" else _backend.F32
    base = torch.randn(n, max(dims), dtype=torch.float64)
    zs = []
    for i, d in enumerate(dims):
        mix = torch.randn(max(dims), d, dtype=torch.float64) / max(dims) ** 0.5
        zs.append((0.6 * base @ mix + torch.randn(n, d, dtype=torch.float64) + 0.3 * i).to(tdt).cuda().contiguous())
    m = len(dims)
    dims_a = (C.c_int64 * m)(*dims)
    nbytes = H.lib.ccz_pair_loss_state_bytes(code, dims_a, m)
    D = sum(dims)
    assert nbytes == (D + 3) * D * 8 + D * D * 4          # Gamma | centring row | batch mean | pilot correction | Gamma fp32
    state = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device="cuda")
    loss = torch.empty((), dtype=tdt, device="cuda")
    torch.cuda.synchronize()
    H.check(H.lib.ccz_pair_loss_forward(H.raw, code, _views(zs), m, n, 1e-4, C.c_void_p(loss.data_ptr()), C.c_void_p(state.data_ptr())))
    scale = torch.tensor(-1.75, dtype=tdt, device="cuda")
    grads = [torch.full_like(z, float("nan")) for z in zs]
    gp = (C.c_void_p * m)(*[g.data_ptr() for g in grads])
    ldg = (C.c_int64 * m)(*[int(g.stride(0)) for g in grads])
    for rep in range(2):                       # the state survives a backward (retain_graph)
        H.check(H.lib.ccz_pair_loss_backward(H.raw, code, _views(zs), m, n, C.c_void_p(state.data_ptr()), C.c_void_p(scale.data_ptr()), gp, ldg))
    H.sync()
    z64 = [z.double().cpu().numpy() for z in zs]
    want_l, want_g = 0.0, [np.zeros_like(z) for z in z64]
    for a in range(m):
        for b in range(a + 1, m):
            l, ga, gb = ol.cca_loss_closed_form(z64[a], z64[b], 1e-4)
            want_l += l
            want_g[a] += ga
            want_g[b] += gb
    assert abs(loss.item() - want_l) <= tol * abs(want_l)
    for g, w in zip(grads, want_g):
        assert rel_err(g.cpu().numpy(), -1.75 * w) < 10 * tol
    # forward only (no state) gives the same value; one-sided backward leaves the other tensor alone
    loss2 = torch.empty((), dtype=tdt, device="cuda")
    H.check(H.lib.ccz_pair_loss_forward(H.raw, code, _views(zs), m, n, 1e-4, C.c_void_p(loss2.data_ptr()), None))
    gp1 = (C.c_void_p * m)(*([grads[0].data_ptr()] + [None] * (m - 1)))
    grads[0].fill_(0.0)
    keep = grads[1].clone()
    H.check(H.lib.ccz_pair_loss_backward(H.raw, code, _views(zs), m, n, C.c_void_p(state.data_ptr()), None, gp1, ldg))
    H.sync()
    assert abs(loss2.item() - want_l) <= tol * abs(want_l)
    assert rel_err(grads[0].cpu().numpy(), want_g[0]) < 10 * tol
    assert torch.equal(keep, grads[1])


def test_module_backward_scales_with_the_upstream_gradient():
    """(3 loss).backward() and a loss inside a sum: the node's backward receives a non-unit grad_out (the reference:
    autograd through CCALoss.forward)."""
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss, MCCALoss
    from oracle import losses as ol

    torch.manual_seed(4)
    z1 = torch.randn(4096, 512, device="cuda", requires_grad=True)
    z2 = (0.5 * z1.detach() + torch.randn(4096, 512, device="cuda")).requires_grad_(True)
    z3 = (0.3 * z1.detach()[:, :256] + torch.randn(4096, 256, device="cuda")).requires_grad_(True)
    obj = CCALoss(eps=1e-4)
    (3.0 * obj([z1, z2]) + 0.5 * MCCALoss(eps=1e-4)([z1, z2, z3])).backward()
    a, b, cc = (t.detach().double().cpu().numpy() for t in (z1, z2, z3))
    _, g12a, g12b = ol.cca_loss_closed_form(a, b, 1e-4)
    _, g13a, g13c = ol.cca_loss_closed_form(a, cc, 1e-4)
    _, g23b, g23c = ol.cca_loss_closed_form(b, cc, 1e-4)
    assert rel_err(z1.grad.cpu().numpy(), 3.5 * g12a + 0.5 * g13a) < 1e-2
    assert rel_err(z2.grad.cpu().numpy(), 3.5 * g12b + 0.5 * g23b) < 1e-2
    assert rel_err(z3.grad.cpu().numpy(), 0.5 * (g13c + g23c)) < 1e-2


def test_fast_path_offset_embeddings_and_row_tails():
    """Post-ReLU style embeddings (means at 20 sigma), a batch that is not a multiple of anything: the pilot comes from
    k_colsum_pilot, the shift is undone in k_loss_prep_partials."""
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss
    from oracle import losses as ol

    torch.manual_seed(8)
    n = 5003
    z1 = (torch.randn(n, 256, dtype=torch.float64) + 20.0)
    z2 = (0.6 * z1[:, :128] + torch.randn(n, 128, dtype=torch.float64) * 0.5 - 11.0)
    a = z1.float().cuda().requires_grad_(True)
    b = z2.float().cuda().requires_grad_(True)
    loss = CCALoss(eps=1e-4)([a, b])
    loss.backward()
    l, g1, g2 = ol.cca_loss_closed_form(a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy(), 1e-4)
    assert abs(loss.item() - l) <= 1e-3 * abs(l)
    assert rel_err(a.grad.cpu().numpy(), g1) < 1e-2 and rel_err(b.grad.cpu().numpy(), g2) < 1e-2
