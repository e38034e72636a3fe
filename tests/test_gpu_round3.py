"""GPU parity, round 3: the DCCA loss at widths beyond the batched step kernels (the metric shape d = 4096), the
stream-native objective contract, many-view losses at any width, and the drifting host-streamed input.

Everything goes through the C ABI (`ccz_pair_loss` / `ccz_cca_loss` / `ccz_moments`); the comparator is
`oracle.losses` (float64 closed form of cca_zoo/deep/objectives.py:61-102,138-153 + autograd) or the reference's own
goldens.
"""

import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, rel_err

pytestmark = pytest.mark.gpu


def _pair(n, d1, d2, seed, dtype, offset=0.0):
    """Correlated batch with a non-trivial spectrum: z2 mixes a projection of z1 with noise; optional common offset."""
    import torch

    g = torch.Generator().manual_seed(seed)
    z1 = torch.randn(n, d1, generator=g, dtype=torch.float64)
    mix = torch.randn(d1, d2, generator=g, dtype=torch.float64) / np.sqrt(d1)
    z2 = 0.7 * z1 @ mix + torch.randn(n, d2, generator=g, dtype=torch.float64)
    z1 = z1 * (0.5 + torch.rand(d1, generator=g, dtype=torch.float64)) + offset
    z2 = z2 + 0.5 * offset
    return z1.to(dtype), z2.to(dtype)


# ---------------------------------------------------------------------------------------------
# the wide route of pair_core (csrc/loss.hip): any view wider than 2048 columns
# ---------------------------------------------------------------------------------------------
_WIDE_CLOSED_FORM = {}


@pytest.mark.parametrize("d1,d2,kind,offset", [
    (2304, 2304, "f64", 0.0), (2560, 2049, "f64", 3.0), (4096, 4096, "f64", 0.0),
    (2304, 2304, "f32", 0.0), (2560, 2049, "f32", 3.0), (4096, 4096, "f32", 0.0),
])
def test_wide_cca_loss_against_closed_form(d1, d2, kind, offset):
    """n = 16384, value + both gradients, 1e-5 (float64) / 1e-3 (float32); odd widths (2049) leave every aligned fast
    path; offset means exercise the pilot decision of the wide route."""
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss, check_async_errors
    from oracle import losses as ol

    n, eps = 16384, 1e-4
    tdt = torch.float64 if kind == "f64" else torch.float32
    tol = 1e-5 if kind == "f64" else 1e-3
    z1, z2 = _pair(n, d1, d2, d1 + d2, tdt, offset)
    a = z1.cuda().requires_grad_(True)
    b = z2.cuda().requires_grad_(True)
    loss = CCALoss(eps=eps)([a, b])
    loss.backward()
    check_async_errors()
    # the host closed form (two dense eigen-solves: 20 s at 4096) is evaluated ONCE per shape, on the float64 batch; the
    # float32 case feeds the kernel that batch rounded to float32, which moves the closed form by ~1e-6 << its 1e-3 bar
    key = (d1, d2, offset)
    if key not in _WIDE_CLOSED_FORM:
        y1, y2 = _pair(n, d1, d2, d1 + d2, torch.float64, offset)
        _WIDE_CLOSED_FORM[key] = ol.cca_loss_closed_form(y1.numpy(), y2.numpy(), eps)
    l, g1, g2 = _WIDE_CLOSED_FORM[key]
    assert abs(loss.item() - l) <= tol * abs(l), (loss.item(), l)
    assert rel_err(a.grad.cpu().numpy(), g1) < tol
    assert rel_err(b.grad.cpu().numpy(), g2) < tol
    if (d1, d2, kind) == (2560, 2049, "f64"):
        # one-sided gradients and forward only on the same route
        a2 = z1.cuda().requires_grad_(True)
        l2 = CCALoss(eps=eps)([a2, z2.cuda()])
        l2.backward()
        assert abs(l2.item() - l) <= tol * abs(l) and rel_err(a2.grad.cpu().numpy(), g1) < tol
        b2 = z2.cuda().requires_grad_(True)
        CCALoss(eps=eps)([z1.cuda(), b2]).backward()
        assert rel_err(b2.grad.cpu().numpy(), g2) < tol
        with torch.no_grad():
            assert abs(CCALoss(eps=eps)([z1.cuda(), z2.cuda()]).item() - l) <= tol * abs(l)


_WIDE_GOLDEN_SCRIPT = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np, torch
from conftest import load_golden, rel_err
from cca_zoo_amd.deep.objectives import CCALoss, MCCALoss, check_async_errors
g = load_golden("losses")
tags = sorted({{k.rsplit("/", 1)[0] for k in g if k.startswith("cca/")}})
assert len(tags) >= 10
for tag in tags:
    eps = 1e-5 if "unequal" in tag else float(tag.split("eps")[1])
    z1 = torch.tensor(g[tag + "/z1"], device="cuda", requires_grad=True)
    z2 = torch.tensor(g[tag + "/z2"], device="cuda", requires_grad=True)
    loss = CCALoss(eps=eps)([z1, z2]); loss.backward()
    f32 = z1.dtype == torch.float32
    ref = float(g[tag + "/loss"])
    assert abs(loss.item() - ref) <= (1e-3 if f32 else 1e-5) * abs(ref), tag
    assert rel_err(z1.grad.cpu().numpy(), g[tag + "/g1"]) < (5e-2 if f32 else 1e-5), tag
    assert rel_err(z2.grad.cpu().numpy(), g[tag + "/g2"]) < (5e-2 if f32 else 1e-5), tag
zs = [torch.tensor(g[f"mcca/z{{i}}"], device="cuda", requires_grad=True) for i in range(3)]
loss = MCCALoss(eps=1e-5)(zs); loss.backward()
assert abs(loss.item() - float(g["mcca/loss"])) < 1e-5 * abs(float(g["mcca/loss"]))
for i in range(3):
    assert rel_err(zs[i].grad.cpu().numpy(), g[f"mcca/g{{i}}"]) < 1e-5
check_async_errors()
print("WIDE-ROUTE-GOLDENS-OK", len(tags))
"""


def test_wide_route_reproduces_the_reference_goldens():
    """CCZ_LOSS_FUSED=0 (read once per process, hence the child process) sends EVERY shape through the wide route of
    pair_core: the reference's own CCALoss / MCCALoss values and gradients (tests/golden/losses.npz) pin it."""
    env = dict(os.environ, CCZ_LOSS_FUSED="0")
    code = _WIDE_GOLDEN_SCRIPT.format(root=ROOT, tests=os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "WIDE-ROUTE-GOLDENS-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_metric_shape_loss_full_size():
    """BASELINE's second metric at its own shape: CCALoss fwd+bwd on n = 1e6, 2 x 4096, float32.  Comparator: float64
    moments of the same float32 batch (chunked on the device by torch -- the test's comparator, not the product), the
    closed form evaluated on the host from those moments (oracle.losses.cca_loss_from_moments), the gradient on 2048
    rows spread over the batch as (Z - mean) Gamma_oracle."""
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss, check_async_errors
    from oracle import losses as ol

    torch.cuda.empty_cache()                                # blocks cached by earlier tests are not "free" to mem_get_info
    free, _ = torch.cuda.mem_get_info()
    n, d, eps = 1_000_000, 4096, 1e-6
    if free < 4.4 * n * d * 4:
        pytest.skip("needs ~72 GB of free HBM")
    torch.manual_seed(5)
    z1 = torch.empty(n, d, device="cuda")
    z2 = torch.empty(n, d, device="cuda")
    lat = 96
    w1 = torch.randn(lat, d, device="cuda") / np.sqrt(lat)
    w2 = torch.randn(lat, d, device="cuda") / np.sqrt(lat)
    chunk = 62500
    for r0 in range(0, n, chunk):
        zl = torch.randn(chunk, lat, device="cuda")
        z1[r0:r0 + chunk] = zl @ w1 + torch.randn(chunk, d, device="cuda") + 0.25
        z2[r0:r0 + chunk] = zl @ w2 + torch.randn(chunk, d, device="cuda")
    z1.requires_grad_(True)
    z2.requires_grad_(True)
    loss = CCALoss(eps=eps)([z1, z2])
    loss.backward()
    check_async_errors()
    # comparator: float64 moments, chunk by chunk
    D = 2 * d
    G = torch.zeros(D, D, dtype=torch.float64, device="cuda")
    s = torch.zeros(D, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        for r0 in range(0, n, 31250):
            blk = torch.cat([z1[r0:r0 + 31250], z2[r0:r0 + 31250]], dim=1).double()
            G += blk.T @ blk
            s += blk.sum(0)
            del blk
    l_ref, Gamma, mean = ol.cca_loss_from_moments(G.cpu().numpy(), s.cpu().numpy(), n, d, d, eps)
    del G
    assert abs(loss.item() - l_ref) <= 1e-3 * abs(l_ref), (loss.item(), l_ref)
    rows = torch.arange(0, n, n // 2048, device="cuda")[:2048]
    with torch.no_grad():
        Z = torch.cat([z1[rows], z2[rows]], dim=1).double().cpu().numpy()
    g_ref = (Z - mean) @ Gamma
    g1 = z1.grad[rows].double().cpu().numpy()
    g2 = z2.grad[rows].double().cpu().numpy()
    assert rel_err(g1, g_ref[:, :d]) < 1e-3 and rel_err(g2, g_ref[:, d:]) < 1e-3


# ---------------------------------------------------------------------------------------------
# MCCALoss: any number of views (<= 8) of any width through ONE fused pass
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dims,kind,n", [
    ((2304, 512, 256), "f64", 8192), ((2100, 2100, 300), "f32", 8192), ((64, 40, 33, 20, 8), "f64", 1000),
    ((256, 256, 256), "f32", 4096),
])
def test_mcca_loss_any_width_against_pairwise_oracle(dims, kind, n):
    import torch

    from cca_zoo_amd.deep.objectives import MCCALoss, check_async_errors
    from oracle import losses as ol

    tdt = torch.float64 if kind == "f64" else torch.float32
    tol = 1e-5 if kind == "f64" else 1e-3
    g = torch.Generator().manual_seed(sum(dims))
    lat = torch.randn(n, 24, generator=g, dtype=torch.float64)
    zs = [(lat @ torch.randn(24, d, generator=g, dtype=torch.float64) / 5.0 + torch.randn(n, d, generator=g, dtype=torch.float64) + 0.3 * i).to(tdt)
          for i, d in enumerate(dims)]
    ts = [z.cuda().requires_grad_(True) for z in zs]
    loss = MCCALoss(eps=1e-4)(ts)
    loss.backward()
    check_async_errors()
    l, grads = ol.mcca_loss_closed_form([z.numpy() for z in zs], 1e-4)
    assert abs(loss.item() - l) <= max(tol, 2e-6) * abs(l)          # fp32 accumulator of the reference (:149)
    for t, gr in zip(ts, grads):
        assert rel_err(t.grad.cpu().numpy(), gr) < tol


def test_pair_loss_abi_argument_checks():
    import ctypes as C

    import torch

    from cca_zoo_amd import _backend

    h = _backend.default_handle(0)
    z = torch.randn(64, 8, device="cuda")
    loss = torch.empty((), device="cuda")
    views = (_backend.View * 9)()
    for i in range(9):
        views[i].data, views[i].cols, views[i].ld = z.data_ptr(), 8, 8
    with pytest.raises(ValueError, match="2 .. 8 views"):
        h.check(h.lib.ccz_pair_loss(h.raw, _backend.F32, views, 9, 64, 1e-4, C.c_void_p(loss.data_ptr()), None, None))
    with pytest.raises(ValueError, match="2 .. 8 views"):
        h.check(h.lib.ccz_pair_loss(h.raw, _backend.F32, views, 1, 64, 1e-4, C.c_void_p(loss.data_ptr()), None, None))
    views[1].ld = 4
    with pytest.raises(ValueError, match="bad shape"):
        h.check(h.lib.ccz_pair_loss(h.raw, _backend.F32, views, 2, 64, 1e-4, C.c_void_p(loss.data_ptr()), None, None))


# ---------------------------------------------------------------------------------------------
# stream-native objective (the loss is an ordinary autograd node of training_step)
# ---------------------------------------------------------------------------------------------
def test_loss_on_a_side_stream_matches_the_default_stream():
    """The wrappers join libccz's stream to torch's CURRENT stream on the device (ccz_stream_acquire / release):
    producers and consumers on a non-default stream see correctly ordered data without any host synchronisation."""
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss, MCCALoss

    torch.manual_seed(3)
    x1 = torch.randn(4096, 96, device="cuda")
    x2 = torch.randn(4096, 80, device="cuda")
    e1, e2 = torch.nn.Linear(96, 64).cuda(), torch.nn.Linear(80, 48).cuda()

    def run(stream):
        for p in list(e1.parameters()) + list(e2.parameters()):
            p.grad = None
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream())
        with ctx:
            # a long producer kernel chain right before the loss, a consumer right after: ordering errors would show
            a = e1(x1)
            for _ in range(20):
                a = a + 1e-3 * torch.tanh(a)
            b = e2(x2)
            loss = CCALoss(eps=1e-4)([a, b]) + 0.1 * MCCALoss(eps=1e-4)([a, b, b * 0.5 + a[:, :48]])
            loss.backward()
            out = (loss.detach().clone(), [p.grad.detach().clone() for p in e1.parameters()])
        if stream is not None:
            stream.synchronize()
        else:
            torch.cuda.synchronize()
        return out

    l0, g0 = run(None)
    side = torch.cuda.Stream()
    for _ in range(3):
        l1, g1 = run(side)
        assert abs(l1.item() - l0.item()) <= 1e-5 * abs(l0.item())
        for p, q in zip(g0, g1):
            assert rel_err(q.cpu().numpy(), p.cpu().numpy()) < 1e-4


def test_non_spd_is_reported_asynchronously():
    """eps = 0 with a duplicated column: the factorization meets a zero pivot.  The call itself does not wait for the
    device: the loss comes back NaN, and the NEXT loss call (or check_async_errors) raises LinAlgError naming the view."""
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss, check_async_errors

    check_async_errors()
    torch.manual_seed(0)
    z1 = torch.randn(512, 16, dtype=torch.float64, device="cuda")
    z2 = torch.randn(512, 12, dtype=torch.float64, device="cuda")
    z2[:, 7] = z2[:, 3]
    loss = CCALoss(eps=0.0)([z1, z2])
    assert torch.isnan(loss).item()
    with pytest.raises(np.linalg.LinAlgError, match="S_22"):
        check_async_errors()
    check_async_errors()                                    # the record is cleared once reported
    CCALoss(eps=0.0)([z1, z2])
    torch.cuda.synchronize()
    with pytest.raises(np.linalg.LinAlgError, match="not positive definite"):
        CCALoss(eps=1e-4)([z1, z2])                         # the NEXT call reports the earlier failure
    assert torch.isfinite(CCALoss(eps=1e-4)([z1, z2])).item()
    check_async_errors()


def test_loss_calls_do_not_synchronise_the_host():
    """Enqueue 40 loss evaluations behind a long-running kernel chain: the host returns from all of them while the
    device is still busy (it would take > 40 x the chain if any call drained the queue)."""
    import time

    import torch

    from cca_zoo_amd.deep.objectives import CCALoss

    torch.manual_seed(0)
    z1 = torch.randn(8192, 256, device="cuda", requires_grad=True)
    z2 = torch.randn(8192, 256, device="cuda", requires_grad=True)
    obj = CCALoss(eps=1e-4)
    obj([z1, z2]).backward()
    big = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        big = torch.tanh(big @ big * 1e-4)                  # ~30 x 7 ms of device work queued ahead
    t_enqueue_chain = time.perf_counter() - t0
    t1 = time.perf_counter()
    for _ in range(10):
        obj([z1, z2]).backward()
    t_enqueue_losses = time.perf_counter() - t1
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # the chain alone needs ~0.2 s on the device; ten enqueue-only losses return in a fraction of that
    assert t_all > 0.12, t_all
    assert t_enqueue_chain + t_enqueue_losses < 0.6 * t_all, (t_enqueue_chain, t_enqueue_losses, t_all)


# ---------------------------------------------------------------------------------------------
# host-streamed fp32 input whose FIRST chunk is centred and whose later rows are far from zero
# ---------------------------------------------------------------------------------------------
def test_streamed_chunks_with_drifting_means(monkeypatch):
    from cca_zoo_amd import _backend
    from cca_zoo_amd._moments import compute_moments

    monkeypatch.setenv("CCZ_H2D_CHUNK_MB", "8")             # 2048-row chunks of the 2 x 512 float32 views below
    H = _backend.default_handle(0)
    rng = np.random.default_rng(11)
    n, d = 32768, 512
    lat = rng.standard_normal((n, 6))
    views = []
    for v in range(2):
        x = lat @ rng.standard_normal((6, d)) + rng.standard_normal((n, d))
        x[4096:] += 100.0 * (1 + v)                          # the first two chunks are centred, the rest sit at 100-200 sigma
        views.append(np.ascontiguousarray(x, dtype=np.float32))
    mom, keep, nt, dims, kind = compute_moments(views, H)
    assert kind == "f32" and H.moments_last_pilot()
    D = 2 * d
    flat = H.to_host(mom, (D * D + D,))
    del keep
    G, s = flat[:D * D].reshape(D, D), flat[D * D:]
    X = np.hstack(views).astype(np.float64)
    np.testing.assert_allclose(s / n, X.mean(axis=0), rtol=1e-9)
    # The drift is a (legitimate) huge between-segment variance that would mask everything on the scale of the total
    # covariance: compare the WITHIN-segment scatter  W = S_total - n_A n_B / n (mu_A - mu_B)(mu_A - mu_B)'  -- raw fp32
    # products of values near 100-200 would miss it by ~5e-3 of its scale
    nA, nB = 4096, n - 4096
    delta = X[:nA].mean(axis=0) - X[nA:].mean(axis=0)
    between = nA * nB / n * np.outer(delta, delta)
    iu = np.triu_indices(D)
    W_dev = G[iu] - (np.outer(s, s) / n)[iu] - between[iu]
    XA, XB = X[:nA] - X[:nA].mean(axis=0), X[nA:] - X[nA:].mean(axis=0)
    W_ref = XA.T @ XA + XB.T @ XB
    scale = np.sqrt(np.outer(np.diag(W_ref), np.diag(W_ref)))[iu]
    assert np.abs((W_dev - W_ref[iu]) / scale).max() < 2e-5


# ---------------------------------------------------------------------------------------------
# pilot shift inside the FIFO kernel (chip-filling grids of data far from zero)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [65536, 65536 + 19, 40000, 300011])
def test_pilot_inside_the_fifo_kernel(n):
    """fp32 views at 100 sigma from zero on a grid large enough for the XCD-sliced FIFO kernel (2 x 4096 columns):
    k_gram_f32_fifo<PILOT> on the rows that form whole 32-row ring periods + the staged kernel on the ragged tail.
    Covariance from the device moments against float64 (torch on the device as the comparator) within 2e-5 of scale;
    raw fp32 products would be off by ~5e-2."""
    import torch

    from cca_zoo_amd import _backend
    from cca_zoo_amd._moments import compute_moments

    H = _backend.default_handle(0)
    d = 4096
    g = torch.Generator(device="cuda").manual_seed(n)
    lat = torch.randn(n, 8, device="cuda", generator=g)
    tv = []
    for v in range(2):
        x = lat @ torch.randn(8, d, device="cuda", generator=g) + torch.randn(n, d, device="cuda", generator=g)
        sd = x.std(dim=0)
        tv.append((x + 100.0 * sd * (1.0 + 0.5 * v)).contiguous())
    mom, keep, nt, dims, kind = compute_moments(tv, H)
    assert kind == "f32" and H.moments_last_pilot()
    D = 2 * d
    mt = keep[0]
    G = mt[: D * D].reshape(D, D)
    s = mt[D * D:]
    X = torch.cat([t.double() for t in tv], dim=1)
    mu = X.mean(0)
    Xc = X - mu
    Cref = Xc.T @ Xc / (n - 1)
    del X, Xc
    Cdev = (torch.triu(G) - torch.triu(torch.outer(s, s)) / n) / (n - 1)
    sc = torch.sqrt(torch.outer(torch.diag(Cref), torch.diag(Cref)))
    err = (torch.triu(Cdev - Cref) / sc).abs().max().item()
    assert err < 2e-5, err
    assert torch.allclose(s / n, mu, rtol=1e-9)
    del keep


# ---------------------------------------------------------------------------------------------
# per-column parity at the metric's dimensions on a WELL-POSED spectrum
# ---------------------------------------------------------------------------------------------
def test_ns_dimensions_per_column_on_a_separated_spectrum():
    """CCA, 2 x 4096 features, k = 64 -- the metric's dimensions -- on data whose 64 leading canonical correlations are
    separated by ~1.3e-2 (population values 0.985, 0.9722, ... 0.1786: orthonormal loadings of strength rho / (1 - rho)
    per latent, unit noise, n = 1e6 so that the sample noise floor ~0.13 stays below the last one; the sample gaps are
    asserted to exceed 8e-3).  With such gaps every single
    direction is well defined and the north-star bar applies PER COLUMN: weights and correlations within 1e-5
    (float64 views) / 1e-3 (float32 views) of oracle.gram_form on the float64 moments of the data
    (cca_zoo/linear/_rcca.py:92-100).  One data set and ONE oracle solve (40 s of host LAPACK) serve both dtypes: the
    float32 fit sees the float64 views rounded to float32, which moves the oracle's answer by ~1e-6 of a column, three
    orders below that fit's 1e-3 bar."""
    import torch

    from cca_zoo_amd.linear import CCA
    from conftest import col_rel_err
    from oracle import gram_form as gf

    d, k, n = 4096, 64, 1_000_000
    torch.cuda.empty_cache()                                # blocks cached by earlier tests are not "free" to mem_get_info
    free, _ = torch.cuda.mem_get_info()
    if free < 2.6 * n * 2 * d * 8 + 12e9:
        pytest.skip("not enough free HBM")
    rho = 0.985 - 0.0128 * np.arange(k)
    g = torch.Generator(device="cuda").manual_seed(77)
    amp = torch.as_tensor(np.sqrt(rho / (1.0 - rho)), dtype=torch.float64, device="cuda")
    loads = []
    for v in range(2):
        q, _ = torch.linalg.qr(torch.randn(d, k, dtype=torch.float64, device="cuda", generator=g))
        loads.append((q * amp).T.contiguous())                      # k x d, row j = sqrt(s_j) q_j'
    tv = [torch.empty(n, d, dtype=torch.float64, device="cuda") for _ in range(2)]
    step = 31250
    for r0 in range(0, n, step):
        z = torch.randn(step, k, dtype=torch.float64, device="cuda", generator=g)
        for v in range(2):
            tv[v][r0:r0 + step] = z @ loads[v] + torch.randn(step, d, dtype=torch.float64, device="cuda", generator=g)
    m = CCA(latent_dimensions=k).fit(tv)
    # comparator: float64 moments of the same data (torch, chunked), oracle solve on the host
    D = 2 * d
    G = torch.zeros(D, D, dtype=torch.float64, device="cuda")
    s = torch.zeros(D, dtype=torch.float64, device="cuda")
    for r0 in range(0, n, step):
        X = torch.cat([t[r0:r0 + step] for t in tv], dim=1)
        G += X.T @ X
        s += X.sum(0)
        del X
    Gh, sh = G.cpu().numpy(), s.cpu().numpy()

    def solve():
        W_, means_, sv_ = gf.rcca_from_moments(Gh, sh, n, [d, d], k, c=[0.0, 0.0], fast=True)
        return {"W0": W_[0], "W1": W_[1], "mean0": means_[0], "mean1": means_[1], "sv": sv_}

    from conftest import host_solve_cached, moments_probe

    o = host_solve_cached("ns_separated_1e6_4096", moments_probe(Gh, sh), solve)      # one oracle solve: 40 s of host LAPACK
    W, means, sv = [o["W0"], o["W1"]], [o["mean0"], o["mean1"]], o["sv"]
    del G, Gh
    gaps = -np.diff(sv)
    assert gaps.min() > 8e-3 and sv[0] < 0.99 and sv[-1] > 0.15, (gaps.min(), sv[0], sv[-1])   # the problem IS well posed

    def check(model, views, tol):
        np.testing.assert_allclose(model.singular_values_, sv, rtol=tol)
        for w, r in zip(model.weights_, W):
            assert w.shape == (d, k)
            assert col_rel_err(w, r) < tol, col_rel_err(w, r)
        np.testing.assert_allclose(model.score(views), sv, atol=10 * tol)

    check(m, tv, 1e-5)
    tv32 = []
    while tv:                                               # float32 views of the same rows, the float64 ones released one by one
        tv32.append(tv.pop(0).float())
    torch.cuda.empty_cache()
    m32 = CCA(latent_dimensions=k).fit(tv32)
    assert m32.weights_[0].dtype == np.float32
    check(m32, tv32, 1e-3)


# ---------------------------------------------------------------------------------------------
# the two-part exchange of the sharded fit on real streams
# ---------------------------------------------------------------------------------------------
def test_blocks_layout_on_the_device():
    import torch

    from cca_zoo_amd import _backend
    from test_round3_host import _blocks_reference

    H = _backend.default_handle(0)
    rng = np.random.default_rng(2)
    for dims in ([300, 129], [64, 200, 65]):
        D = sum(dims)
        X = rng.standard_normal((50, D))
        G, s = np.triu(X.T @ X), X.sum(0)
        mom = torch.as_tensor(np.concatenate([G.ravel(), s]), device="cuda")
        head_ref, tail_ref = _blocks_reference(G, s, dims)
        n_head = head_ref.size
        packed = torch.zeros(n_head + tail_ref.size, dtype=torch.float64, device="cuda")
        H.moments_pack_blocks(mom.data_ptr(), D, dims, packed.data_ptr())
        H.sync()
        got = packed.cpu().numpy()
        assert np.array_equal(got[:n_head - 1], head_ref[:-1]) and np.array_equal(got[n_head:], tail_ref)
        out = torch.zeros_like(mom)
        side = torch.cuda.Stream()
        H.moments_unpack_blocks(packed.data_ptr(), D, dims, out.data_ptr(), H.HEAD)
        H.moments_unpack_blocks(packed.data_ptr(), D, dims, out.data_ptr(), H.TAIL, on_stream=side.cuda_stream)
        H.sync()
        side.synchronize()
        o = out.cpu().numpy()
        assert np.array_equal(np.triu(o[:D * D].reshape(D, D)), G) and np.array_equal(o[D * D:], s)


def test_sharded_fits_world_size_one_use_the_deferred_exchange():
    """rCCA / MCCA / GCCA inside row_sharded() (RCCL, world size 1): the two-part exchange with the off-diagonal part
    unpacked on a side stream and awaited by the solve on the device gives the unsharded fit."""
    import torch
    import torch.distributed as dist

    from cca_zoo_amd import row_sharded
    from cca_zoo_amd.linear import GCCA, MCCA, rCCA
    from conftest import col_rel_err

    started = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29671", rank=0, world_size=1)
        started = True
    try:
        torch.manual_seed(1)
        n = 20000
        lat = torch.randn(n, 6, device="cuda", dtype=torch.float64) * torch.linspace(2.0, 0.6, 6, device="cuda", dtype=torch.float64)
        views = [lat @ torch.randn(6, d, device="cuda", dtype=torch.float64) + torch.randn(n, d, device="cuda", dtype=torch.float64)
                 for d in (1100, 1280, 300)]
        for make, vs in ((lambda: rCCA(latent_dimensions=5, c=0.05), views[:2]), (lambda: MCCA(latent_dimensions=5, c=0.1), views),
                         (lambda: GCCA(latent_dimensions=5, c=0.1), views)):
            ref = make().fit(vs)
            for _ in range(3):                                # repeated: the exchange buffer and the side stream are reused
                with row_sharded():
                    m = make().fit(vs)
                assert m.n_samples_ == n
                for a, b in zip(m.weights_, ref.weights_):
                    assert col_rel_err(a, b) < 1e-9
    finally:
        if started:
            dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
# A B' fp64 products at the shapes of the Cholesky updates and forward triangular solves (csrc/gemm64_big.hip)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,lda_extra,beta", [
    (4096, 512, 512, 0, 0.0),        # a super-block product: fewer 128-tiles than CUs -> the 64 x 128 tile
    (4096, 3584, 512, 512, 1.0),     # a trailing update inside a wider matrix (strided operands)
    (1000, 777, 96, 2, -0.5),        # ragged rows and columns
    (130, 129, 32, 0, 0.0),          # barely two tiles each way
    (2048, 2048, 4096, 0, 1.0),
])
def test_gemm_f64_a_bt_shapes(M, N, K, lda_extra, beta):
    import ctypes as C

    import torch

    from cca_zoo_amd import _backend

    H = _backend.default_handle(0)
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K + lda_extra, dtype=torch.float64, device="cuda")
    B = torch.randn(N, K + lda_extra, dtype=torch.float64, device="cuda")
    Cm = torch.randn(M, N + 2, dtype=torch.float64, device="cuda")
    ref = Cm.clone()
    ref[:, :N] = 0.7 * (A[:, :K] @ B[:, :K].T) + beta * Cm[:, :N]
    H.check(H.lib.ccz_gemm_f64(H.raw, 0, 1, M, N, K, 0.7, C.c_void_p(A.data_ptr()), A.shape[1], C.c_void_p(B.data_ptr()),
                               B.shape[1], beta, C.c_void_p(Cm.data_ptr()), Cm.shape[1]))
    H.sync()
    assert float((Cm[:, :N] - ref[:, :N]).abs().max() / ref[:, :N].abs().max()) < 1e-12
    assert torch.equal(Cm[:, N:], ref[:, N:])                # columns past N untouched
