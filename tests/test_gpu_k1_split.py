"""GPU parity of the split-bf16 K1 route (include/ccz.h: ccz_k1_route, csrc/gram_split.hip) through the C ABI: second
moments of fp32 views from two bf16 planes against float64 NumPy, next to the fp32 MFMA route on the same rows; the
estimators on top of it against the oracle.  Reference arithmetic to stay at or above: float32
(cca_zoo/_utils/_linalg.py:28, cca_zoo/linear/_rcca.py:96)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from cca_zoo_amd import _backend

    h = _backend.default_handle(0)
    yield h
    h.k1_route("auto")


def _moments(H, views, route, accumulate_from=None, on_device=True):
    from cca_zoo_amd import _backend

    n = views[0].shape[0]
    D = sum(v.shape[1] for v in views)
    mom = accumulate_from if accumulate_from is not None else H.alloc((D * D + D) * 8)
    keep, descr = [], []
    for v in views:
        if on_device:
            b = H.to_device(np.ascontiguousarray(v))
            keep.append(b)
            descr.append((b.ptr, v.shape[1], v.shape[1]))
        else:
            descr.append((v, v.shape[1], v.strides[0] // v.itemsize))
    prev = H.k1_route(route)
    try:
        H.moments(descr, n, _backend.F32, on_device, mom.ptr, accumulate=accumulate_from is not None)
        taken = H.moments_last_route()[0]
    finally:
        H.k1_route(prev)
    flat = H.to_host(mom, (D * D + D,))
    return flat[: D * D].reshape(D, D), flat[D * D:], taken, mom


def _ref(views):
    X = np.hstack([v.astype(np.float64) for v in views])
    return X.T @ X, X.sum(axis=0)


def _rel(G, Gr):
    iu = np.triu_indices(G.shape[0])
    scale = np.sqrt(np.outer(np.diag(Gr), np.diag(Gr)))
    return float((np.abs(G - Gr) / scale)[iu].max())


def _latent(n, dims, seed, shift=0.0, k=6):
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, k))
    return [(z @ rng.standard_normal((k, d)) + rng.standard_normal((n, d)) + shift).astype(np.float32) for d in dims]


def test_route_setting_round_trip(H):
    assert H.k1_route(None) == "auto"
    assert H.k1_route("bf16x2") == "auto"
    assert H.k1_route(None) == "bf16x2"
    assert H.k1_route("fp32") == "bf16x2"
    assert H.k1_route("auto") == "fp32"
    from cca_zoo_amd._backend import CCZError

    with pytest.raises((ValueError, CCZError)):
        H.k1_route(7)


@pytest.mark.parametrize("n,dims,shift", [
    (8192, [512, 512], 0.0),          # whole panels, whole k-steps
    (5000, [300, 520], 0.0),          # ragged panels (300 = 256 + 44), row tail inside a k-step
    (4099, [257, 63], 0.0),           # one-column panel remainder, 3 rows in the last k-step
    (20000, [256, 256], 10.0),        # every column mean 10 sigma away from zero: the pilot shift
    (33000, [128, 384, 200], 0.5),    # three views, more than two row chunks
    (40000, [1000], 0.0),             # one view (svd_whiten seam)
])
def test_split_route_matches_float64(H, n, dims, shift):
    views = _latent(n, dims, seed=n + len(dims), shift=shift)
    Gr, sr = _ref(views)
    G2, s2, taken2, _ = _moments(H, views, "bf16x2")
    G1, s1, taken1, _ = _moments(H, views, "fp32")
    assert (taken1, taken2) == ("fp32", "bf16x2")
    e1, e2 = _rel(G1, Gr), _rel(G2, Gr)
    # float32 bar of the path (BASELINE north_star: 1e-3 on weights); K1 itself is held to a few 1e-6 like the fp32 kernel
    assert e2 < 2e-6, (e1, e2)
    np.testing.assert_allclose(s2, sr, rtol=1e-12, atol=1e-7)
    np.testing.assert_allclose(s1, s2, rtol=1e-13, atol=1e-9)    # the column sums do not depend on the route (the same fp64 pass)


def test_split_route_error_not_above_fp32_route_on_long_inputs(H):
    """The gate of the route: on inputs long enough for auto mode to choose it, its error against
    float64 moments is not above the fp32 kernel's on the same rows."""
    views = _latent(131072, [512, 512], seed=5)
    Gr, _ = _ref(views)
    G2, _, _, _ = _moments(H, views, "bf16x2")
    G1, _, _, _ = _moments(H, views, "fp32")
    assert _rel(G2, Gr) <= 1.05 * _rel(G1, Gr), (_rel(G1, Gr), _rel(G2, Gr))


def test_auto_route_thresholds(H):
    import os

    if os.environ.get("CCZ_K1_ROUTE"):
        pytest.skip("CCZ_K1_ROUTE overrides the automatic choice in this process")
    small = _latent(3000, [256, 256], seed=1)
    _, _, taken, _ = _moments(H, small, "auto")
    assert taken == "fp32"                                   # below the route's pay-off: the fp32 kernel's single launch
    big = _latent(65536, [1024, 512], seed=2)
    _, _, taken, _ = _moments(H, big, "auto")
    assert taken == "bf16x2"
    from cca_zoo_amd import _backend

    v64 = [v.astype(np.float64) for v in small]
    mom = H.alloc((512 * 512 + 512) * 8)
    H.k1_route("bf16x2")
    try:
        bufs = [H.to_device(v) for v in v64]
        H.moments([(b.ptr, 256, 256) for b in bufs], 3000, _backend.F64, True, mom.ptr)
        assert H.moments_last_route()[0] == "fp64"           # float64 views never leave the float64 pipe
    finally:
        H.k1_route("auto")


def test_split_route_row_super_chunks_and_accumulate(H, monkeypatch):
    """A scratch budget of a few MB forces several row super-chunks (planes + partial tiles re-used), 256-row chunks per
    workgroup; a second call accumulates into the same moments."""
    monkeypatch.setenv("CCZ_SPLIT_SCRATCH_GB", "0.008")
    monkeypatch.setenv("CCZ_SPLIT_ROWS", "256")
    a = _latent(6000, [300, 200], seed=11)
    b = _latent(2500, [300, 200], seed=12, shift=3.0)
    Ga, sa = _ref(a)
    Gb, sb = _ref(b)
    G, s, taken, mom = _moments(H, a, "bf16x2")
    assert taken == "bf16x2" and _rel(G, Ga) < 2e-6
    G, s, taken, _ = _moments(H, b, "bf16x2", accumulate_from=mom)
    assert _rel(G, Ga + Gb) < 2e-6
    np.testing.assert_allclose(s, sa + sb, rtol=1e-12, atol=1e-7)


@pytest.mark.parametrize("cus", ["64", "0"])
@pytest.mark.parametrize("scratch_gb", [None, "0.02"])
def test_split_route_piped_launch_is_opt_in_and_matches(H, monkeypatch, cus, scratch_gb):
    """CCZ_SPLIT_PIPE (an A/B switch, off by default: DESIGN section 7 item 2): the launch cut into row pieces, the split pass of the
    next piece on the (CU-masked or plain) side stream under the MFMA kernel of the current one -- same moments and column sums as the
    one-piece launch, also across row super-chunks."""
    monkeypatch.setenv("CCZ_SPLIT_ROWS", "256")
    if scratch_gb:
        monkeypatch.setenv("CCZ_SPLIT_SCRATCH_GB", scratch_gb)
    views = _latent(9000, [300, 200], seed=31, shift=2.0)
    Gr, sr = _ref(views)
    G1, s1, taken, _ = _moments(H, views, "bf16x2")
    assert taken == "bf16x2" and _rel(G1, Gr) < 2e-6
    monkeypatch.setenv("CCZ_SPLIT_PIPE", "0.1,0.3")
    monkeypatch.setenv("CCZ_SPLIT_PIPE_CUS", cus)
    G2, s2, taken, _ = _moments(H, views, "bf16x2")
    assert taken == "bf16x2" and _rel(G2, Gr) < 2e-6
    iu = np.triu_indices(G1.shape[0])
    np.testing.assert_allclose(G2[iu], G1[iu], rtol=0, atol=2e-6 * np.abs(np.diag(Gr)).max())
    np.testing.assert_allclose(s2, sr, rtol=1e-12, atol=1e-7)


def test_split_route_host_views_streamed(H, monkeypatch):
    """Pageable host inputs: every streamed chunk takes the split route (its own pilot), strided view included."""
    monkeypatch.setenv("CCZ_H2D_CHUNK_MB", "8")
    n = 40000
    views = _latent(n, [384, 256], seed=21, shift=1.0)
    wide = np.zeros((n, 300), dtype=np.float32)
    wide[:, :256] = views[1]
    views[1] = wide[:, :256]                                 # leading dimension 300
    Gr, sr = _ref(views)
    G, s, taken, _ = _moments(H, views, "bf16x2", on_device=False)
    assert taken == "bf16x2"
    assert _rel(G, Gr) < 2e-6
    np.testing.assert_allclose(s, sr, rtol=1e-12, atol=1e-7)


def test_nonfinite_input_reaches_the_moments(H):
    views = _latent(4096, [256, 256], seed=3)
    views[0][17, 5] = np.inf
    G, s, _, _ = _moments(H, views, "bf16x2")
    assert not np.isfinite(s[5])                             # what compute_moments turns into the ValueError of the estimators


@pytest.mark.parametrize("est", ["rcca", "mcca", "gcca"])
def test_estimators_on_split_route_match_oracle(H, est):
    """configs[1]-like widths with the route forced: weights / correlations against the
    oracle's float64 restatement at the path's float32 bar (1e-3), sign-aligned, separated spectrum."""
    import torch

    from conftest import col_rel_err
    from cca_zoo_amd.linear import GCCA, MCCA, rCCA
    from oracle import reference_form as rf

    n, k = (4096 if est == "gcca" else 40000), 8          # (the oracle's GCCA forms the n x n matrix of the reference)
    rng = np.random.default_rng(7)
    z = rng.standard_normal((n, k)) * np.linspace(2.0, 0.5, k)
    dims = [320, 256] if est == "rcca" else [256, 192, 128]
    views = [(z @ rng.standard_normal((k, d)) + rng.standard_normal((n, d))).astype(np.float32) for d in dims]
    tv = [torch.from_numpy(v).cuda() for v in views]
    prev = H.k1_route("bf16x2")
    try:
        if est == "rcca":
            model = rCCA(latent_dimensions=k, c=0.1).fit(tv)
            W_ref, _ = rf.rcca_weights([v.astype(np.float64) for v in views], k, c=0.1)
        elif est == "mcca":
            model = MCCA(latent_dimensions=k, c=0.1).fit(tv)
            W_ref = rf.mcca_weights([v.astype(np.float64) for v in views], k, c=0.1)[0]
        else:
            model = GCCA(latent_dimensions=k, c=0.1).fit(tv)
            W_ref = rf.gcca_weights([v.astype(np.float64) for v in views], k, c=0.1)[0]
        assert H.moments_last_route()[0] == "bf16x2"
    finally:
        H.k1_route(prev)
    for w, r in zip(model.weights_, W_ref):
        assert col_rel_err(np.asarray(w), r) < 1e-3


def test_loss_backward_on_the_split_route_matches_closed_form(H, monkeypatch):
    """The two-view loss whose backward is a large product (gemm_split.hip): value and gradients against the oracle's closed
    form at the float32 bar, and against the fp32-route gradients of the same batch (CCZ_SPLIT_MIN_FLOP lowered so that a
    batch the oracle finishes in seconds takes the route).  Reference: cca_zoo/deep/objectives.py:61-102."""
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss
    from oracle import losses as ol

    monkeypatch.setenv("CCZ_SPLIT_MIN_FLOP", "1e9")
    n, d = 40000, 384
    torch.manual_seed(3)
    z1 = (torch.randn(n, d, device="cuda") + 1.5).requires_grad_(True)           # off-centre, like post-activation embeddings
    z2 = (0.5 * z1.detach() + torch.randn(n, d, device="cuda")).requires_grad_(True)
    grads = {}
    for route in ("bf16x2", "fp32"):
        prev = H.k1_route(route)
        try:
            z1.grad = z2.grad = None
            loss = CCALoss(eps=1e-4)([z1, z2])
            (3.0 * loss).backward()
            torch.cuda.synchronize()
            grads[route] = (float(loss), z1.grad.cpu().numpy().copy(), z2.grad.cpu().numpy().copy())
        finally:
            H.k1_route(prev)
    l_ref, g1, g2 = ol.cca_loss_closed_form(z1.detach().cpu().numpy(), z2.detach().cpu().numpy(), 1e-4)
    for route, (l, a, b) in grads.items():
        assert abs(l - l_ref) < 1e-3 * abs(l_ref), (route, l, l_ref)
        assert np.linalg.norm(a - 3.0 * g1) < 1e-3 * np.linalg.norm(3.0 * g1), route
        assert np.linalg.norm(b - 3.0 * g2) < 1e-3 * np.linalg.norm(3.0 * g2), route
    # the two routes agree far inside the bar
    assert np.linalg.norm(grads["bf16x2"][1] - grads["fp32"][1]) < 2e-5 * np.linalg.norm(grads["fp32"][1])


@pytest.mark.parametrize("planes", ["2", "3"])
def test_projection_on_the_split_arithmetic_is_opt_in_and_matches_float64(H, monkeypatch, planes):
    """``ccz_transform`` with the handle's route set to bf16x2 EXPLICITLY (csrc/project_split.hip): (X - mean) W against a
    float64 product, ragged row count, k < 64, off-centre data; ``auto`` keeps the fp32 kernel for projections (their error is
    not averaged over the rows like K1's).  Reference: cca_zoo/_base.py:108-123."""
    import ctypes as C

    import torch

    from cca_zoo_amd import _backend

    monkeypatch.setenv("CCZ_PROJECT_PLANES", planes)
    n, d, k = 40003, 2048, 24
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randn(n, d, device="cuda", generator=g) * 1.5 + 3.0
    mean = X.double().mean(0)
    W = torch.randn(d, k, dtype=torch.float64, device="cuda", generator=g) / d ** 0.5
    ref = (X.double() - mean) @ W
    res = {}
    for route in ("auto", "bf16x2"):
        out = torch.full((n, k), float("nan"), device="cuda")
        prev = H.k1_route(route)
        try:
            torch.cuda.synchronize()
            H.check(H.lib.ccz_transform(H.raw, _backend.F32, C.c_void_p(X.data_ptr()), n, d, X.stride(0), C.c_void_p(mean.data_ptr()),
                                        C.c_void_p(W.data_ptr()), k, C.c_void_p(out.data_ptr()), out.stride(0)))
            H.sync()
        finally:
            H.k1_route(prev)
        res[route] = float((out.double() - ref).norm() / ref.norm())
        assert torch.isfinite(out).all()
    assert res["auto"] < 5e-6 and res["bf16x2"] < 2e-5, res
    assert res["bf16x2"] != res["auto"]                      # the opt-in route really ran


def test_pool_trim_returns_the_split_routes_scratch(H):
    """The planes and partial tiles of the split route stay pooled with the handle between fits; ``ccz_pool_trim`` hands every
    unused block back to the driver (and the next launch simply allocates again)."""
    views = _latent(40000, [512, 256], seed=9)
    Gr, _ = _ref(views)
    G, _, taken, _ = _moments(H, views, "bf16x2")
    assert taken == "bf16x2"
    freed = H.pool_trim()
    assert freed >= 40000 * 768 * 4                              # at least the two planes of this launch
    assert H.pool_trim() == 0
    G2, _, _, _ = _moments(H, views, "bf16x2")
    assert _rel(G2, Gr) < 2e-6


def test_loss_reports_the_routes_it_took(H):
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss

    z1 = torch.randn(4096, 64, device="cuda", requires_grad=True)
    z2 = torch.randn(4096, 64, device="cuda", requires_grad=True)
    CCALoss(eps=1e-3)([z1, z2]).backward()
    torch.cuda.synchronize()
    fwd, bwd = H.loss_last_route()
    assert bwd == "fp32"                                          # a small batch keeps the fp32 product (and the fast K1 path)
    assert fwd in ("fp32", "none", "bf16x2")


def test_split_loss_for_small_batches_is_opt_in(H, monkeypatch):
    """BASELINE configs[3] (batch 8192, 2 x 512): by default both products of the loss stay on the fp32 pipe (inside a training step
    the bf16 MFMA bursts cost the neighbouring GEMMs more clock than they save: profiles/r06_loss_c4.md); CCZ_LOSS_K1_SPLIT=2 /
    CCZ_LOSS_BWD_SPLIT=2 opt in, and the two forms agree far inside the 1e-3 bar."""
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss

    torch.manual_seed(0)
    z1 = torch.randn(8192, 512, device="cuda", requires_grad=True)
    z2 = (0.5 * z1.detach() + torch.randn(8192, 512, device="cuda")).requires_grad_(True)

    def run():
        z1.grad = z2.grad = None
        loss = CCALoss(eps=1e-6)([z1, z2])
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), z1.grad.clone(), z2.grad.clone(), H.loss_last_route()

    l0, a0, b0, r0 = run()
    assert r0 == ("fp32", "fp32")
    monkeypatch.setenv("CCZ_LOSS_K1_SPLIT", "2")
    monkeypatch.setenv("CCZ_LOSS_BWD_SPLIT", "2")
    l1, a1, b1, r1 = run()
    assert r1 == ("bf16x2", "bf16x2")
    assert abs(l1 - l0) < 1e-5 * abs(l0)
    assert float((a1 - a0).norm() / a0.norm()) < 5e-5 and float((b1 - b0).norm() / b0.norm()) < 5e-5


def test_split_route_random_shapes(H):
    """Seeded random sweep over row counts, view counts, ragged widths, leading dimensions and offsets (the padding, the k-step
    tails, the panel tables per shape, the row-chunk planner): every case against float64 NumPy."""
    rng = np.random.default_rng(2026)
    for case in range(14):
        n = int(rng.choice([17, 240, 1000, 4097, 9000, 20011, 50000]))
        m = int(rng.integers(1, 5))
        dims = [int(rng.choice([5, 64, 100, 255, 256, 257, 300, 512, 700])) for _ in range(m)]
        shift = float(rng.choice([0.0, 0.0, 2.0, -30.0]))
        views = _latent(n, dims, seed=100 + case, shift=shift)
        if case % 3 == 1:                                       # a strided first view on the host path
            wide = np.zeros((n, dims[0] + 5), dtype=np.float32)
            wide[:, 2:2 + dims[0]] = views[0]
            views[0] = wide[:, 2:2 + dims[0]]
            on_device = False
        else:
            on_device = True
        Gr, sr = _ref(views)
        G, s, taken, _ = _moments(H, views, "bf16x2", on_device=on_device)
        assert taken == "bf16x2"
        tol = 1.2e-5 if n < 4096 else 2e-6                       # (a few dozen rows: the dropped 2^-16 terms do not average out yet)
        assert _rel(G, Gr) < tol, (case, n, dims, shift, _rel(G, Gr))
        np.testing.assert_allclose(s, sr, rtol=1e-11, atol=1e-6)
