"""GPU parity: the device ops of libccz against NumPy (called through the C ABI)."""

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


@pytest.mark.parametrize("tA,tB", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (70, 45, 33), (257, 130, 300), (5, 3, 2), (1, 50, 40)])
def test_gemm_f64(H, tA, tB, M, N, K):
    rng = np.random.default_rng(M * 1000 + N * 10 + K)
    A = rng.standard_normal((K, M) if tA else (M, K))
    B = rng.standard_normal((N, K) if tB else (K, N))
    Cm = rng.standard_normal((M, N))
    ref = 0.7 * (A.T if tA else A) @ (B.T if tB else B) - 1.3 * Cm
    Ad, Bd, Cd = H.to_device(A), H.to_device(B), H.to_device(Cm)
    call(H, "ccz_gemm_f64", tA, tB, M, N, K, 0.7, vp(Ad), A.shape[1], vp(Bd), B.shape[1], -1.3, vp(Cd), N)
    out = H.to_host(Cd, (M, N))
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12 * K)


@pytest.mark.parametrize("tA,tB", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K,beta", [(1111, 999, 16, 0.0), (1111, 999, 48, -1.3), (1400, 1290, 144, 0.5), (4001, 515, 512, 0.0),
                                        (130, 4098, 64, 1.0), (1024, 768, 2048, -0.25)])
def test_gemm_f64_large_tiles(H, tA, tB, M, N, K, beta):
    """The solver's 128- / 64-row-tile kernel (gemm64_big.hip: k_gemm_f64_pipe, software-pipelined k-blocks): ragged edges in both
    dimensions, one / three / many k-blocks, all four transposition pairs, accumulate, and the deep-K few-tile shape that runs
    split-K with the atomic epilogue."""
    rng = np.random.default_rng(M + 3 * N + 7 * K + 2 * tA + tB)
    A = rng.standard_normal((K, M) if tA else (M, K))
    B = rng.standard_normal((N, K) if tB else (K, N))
    Cm = rng.standard_normal((M, N))
    ref = 0.7 * (A.T if tA else A) @ (B.T if tB else B) + beta * Cm
    Ad, Bd, Cd = H.to_device(A), H.to_device(B), H.to_device(Cm)
    call(H, "ccz_gemm_f64", tA, tB, M, N, K, 0.7, vp(Ad), A.shape[1], vp(Bd), B.shape[1], beta, vp(Cd), N)
    out = H.to_host(Cd, (M, N))
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12 * K)


@pytest.mark.parametrize("tA", [0, 1])
@pytest.mark.parametrize("M,N,K,beta", [(2048, 160, 2048, 0.0), (1100, 77, 1024, -1.3), (1537, 192, 4096, 0.5),
                                        (1024, 49, 1040, 0.0), (4096, 80, 8192, 1.0)])
def test_gemm_f64_tall_skinny(H, tA, M, N, K, beta):
    """Stripe kernel of the subspace-iteration applies (gemm64_skinny.hip), incl. split-K and ragged M / N."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M) if tA else (M, K))
    B = rng.standard_normal((K, N))
    Cm = rng.standard_normal((M, N))
    ref = 0.7 * (A.T if tA else A) @ B + beta * Cm
    Ad, Bd, Cd = H.to_device(A), H.to_device(B), H.to_device(Cm)
    call(H, "ccz_gemm_f64", tA, 0, M, N, K, 0.7, vp(Ad), A.shape[1], vp(Bd), N, beta, vp(Cd), N)
    out = H.to_host(Cd, (M, N))
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-12 * K)


def test_gemm_strided_views(H):
    rng = np.random.default_rng(0)
    big = rng.standard_normal((40, 50))
    A = big[3:23, 5:17]             # 20 x 12, lda 50
    B = rng.standard_normal((12, 9))
    out = np.zeros((20, 30))
    Ad, Bd, Cd = H.to_device(big), H.to_device(B), H.to_device(out)
    call(H, "ccz_gemm_f64", 0, 0, 20, 9, 12, 1.0, C.c_void_p(Ad.ptr + (3 * 50 + 5) * 8), 50, vp(Bd), 9, 0.0,
         C.c_void_p(Cd.ptr + 4 * 8), 30)
    got = H.to_host(Cd, (20, 30))
    np.testing.assert_allclose(got[:, 4:13], A @ B, rtol=1e-12, atol=1e-12)
    assert np.all(got[:, :4] == 0) and np.all(got[:, 13:] == 0)


@pytest.mark.parametrize("d", [1, 7, 64, 65, 200, 513])
def test_potrf_and_trsm(H, d):
    rng = np.random.default_rng(d)
    X = rng.standard_normal((d + 10, d))
    S = X.T @ X / d + 0.1 * np.eye(d)
    Sd = H.to_device(S)
    call(H, "ccz_potrf_lower", vp(Sd), d, d)
    L = np.tril(H.to_host(Sd, (d, d)))
    np.testing.assert_allclose(L @ L.T, S, rtol=1e-11, atol=1e-11)
    R = rng.standard_normal((37, d))
    Ld = H.to_device(L)
    for trans in (1, 0):
        Rd = H.to_device(R)
        call(H, "ccz_trsm_right_lower", trans, 37, d, vp(Ld), d, vp(Rd), d)
        got = H.to_host(Rd, (37, d))
        ref = np.linalg.solve(L, R.T).T if trans else np.linalg.solve(L.T, R.T).T
        np.testing.assert_allclose(got, ref, rtol=1e-8, atol=1e-9)


def test_potrf_reports_not_spd(H):
    A = np.eye(70)
    A[66, 66] = -1.0
    Ad = H.to_device(A)
    with pytest.raises(np.linalg.LinAlgError, match="pivot 66"):
        call(H, "ccz_potrf_lower", vp(Ad), 70, 70)


@pytest.mark.parametrize("d", [2, 3, 12, 65, 79, 80, 89, 96, 97, 127, 150, 159, 160, 161, 200])
def test_syevj(H, d):
    rng = np.random.default_rng(d)
    if d == 12:   # +/- eigenvalue pairs
        T = rng.standard_normal((7, 5))
        A = np.block([[np.zeros((7, 7)), T], [T.T, np.zeros((5, 5))]])
    else:
        A = rng.standard_normal((d, d))
        A = A + A.T
    Ad, wd, Vd = H.to_device(A), H.alloc(d * 8), H.alloc(d * d * 8)
    sw = C.c_int(0)
    call(H, "ccz_syevj", vp(Ad), d, vp(wd), vp(Vd), C.byref(sw))
    w, V = H.to_host(wd, (d,)), H.to_host(Vd, (d, d))
    np.testing.assert_allclose(w, np.linalg.eigvalsh(A)[::-1], atol=1e-11 * max(1, np.abs(A).sum(1).max()))
    np.testing.assert_allclose(V @ V.T, np.eye(d), atol=1e-12)
    np.testing.assert_allclose(V @ A @ V.T, np.diag(w), atol=1e-10 * max(1, np.abs(A).sum(1).max()))
    assert 1 <= sw.value <= 40


@pytest.mark.parametrize("shape", [(9, 14), (14, 9), (33, 33), (300, 40)])
def test_gesvj(H, shape):
    rng = np.random.default_rng(1)
    A = rng.standard_normal(shape)
    p, q = shape
    r = min(p, q)
    Ad, Ud, sd, Vd = H.to_device(A), H.alloc(p * r * 8), H.alloc(r * 8), H.alloc(r * q * 8)
    call(H, "ccz_gesvj", vp(Ad), p, q, vp(Ud), vp(sd), vp(Vd), None)
    U, s, Vt = H.to_host(Ud, (p, r)), H.to_host(sd, (r,)), H.to_host(Vd, (r, q))
    np.testing.assert_allclose(s, np.linalg.svd(A, compute_uv=False), atol=1e-11)
    np.testing.assert_allclose(U * s @ Vt, A, atol=1e-11)


@pytest.mark.parametrize("p,k", [(60, 5), (300, 12), (700, 40)])
def test_gevp_topk(H, p, k):
    import scipy.linalg

    rng = np.random.default_rng(2)
    lam = np.concatenate([np.linspace(5.0, 2.0, k), rng.uniform(-3.0, 1.0, p - k)])
    Qm, _ = np.linalg.qr(rng.standard_normal((p, p)))
    A = (Qm * lam) @ Qm.T
    A = 0.5 * (A + A.T)
    Ad, wd, Vd = H.to_device(A), H.alloc(k * 8), H.alloc(p * k * 8)
    call(H, "ccz_gevp_topk", vp(Ad), None, p, k, vp(wd), vp(Vd))
    w, V = H.to_host(wd, (k,)), H.to_host(Vd, (p, k))
    np.testing.assert_allclose(w, np.sort(lam)[::-1][:k], atol=1e-9)
    assert np.linalg.norm(A @ V - V * w) < 1e-8 * np.linalg.norm(A)
    Bm = rng.standard_normal((p, p))
    Bm = Bm @ Bm.T / p + np.eye(p)
    wr, Vr = scipy.linalg.eigh(A, Bm, subset_by_index=[p - k, p - 1])
    Bd = H.to_device(Bm)
    call(H, "ccz_gevp_topk", vp(Ad), vp(Bd), p, k, vp(wd), vp(Vd))
    w, V = H.to_host(wd, (k,)), H.to_host(Vd, (p, k))
    np.testing.assert_allclose(w, wr[::-1], atol=1e-9)
    np.testing.assert_allclose(np.diag(V.T @ Bm @ V), 1.0, atol=1e-9)
    S = np.sign(np.sum(V * Vr[:, ::-1], axis=0))
    assert np.linalg.norm(V * S - Vr[:, ::-1]) < 1e-6 * np.linalg.norm(Vr)


@pytest.mark.parametrize("p,q,k", [(40, 30, 4), (250, 320, 10), (640, 500, 24)])
def test_svd_topk(H, p, q, k):
    rng = np.random.default_rng(3)
    r = min(p, q)
    sv = np.concatenate([np.linspace(3.0, 1.5, k), rng.uniform(0.0, 1.0, r - k)])
    U0, _ = np.linalg.qr(rng.standard_normal((p, r)))
    V0, _ = np.linalg.qr(rng.standard_normal((q, r)))
    T = (U0 * sv) @ V0.T
    Td, Ud, sd, Vd = H.to_device(T), H.alloc(p * k * 8), H.alloc(k * 8), H.alloc(q * k * 8)
    call(H, "ccz_svd_topk", vp(Td), p, q, k, vp(Ud), vp(sd), vp(Vd))
    U, s, V = H.to_host(Ud, (p, k)), H.to_host(sd, (k,)), H.to_host(Vd, (q, k))
    np.testing.assert_allclose(s, np.sort(sv)[::-1][:k], atol=1e-9)
    np.testing.assert_allclose(T @ V, U * s, atol=1e-8)
    np.testing.assert_allclose(U.T @ U, np.eye(k), atol=1e-9)


def test_whitener_inv_sqrtm_against_goldens(H):
    from conftest import col_rel_err, load_golden

    g = load_golden("linalg_seams")
    X = g["X"]
    n, d = X.shape
    Gd = H.to_device(np.ascontiguousarray(X.T @ X))
    for c in (0.0, 0.25, 1.0):
        Wd, ld, r = H.alloc(d * d * 8), H.alloc(d * 8), C.c_int64(0)
        call(H, "ccz_whitener", vp(Gd), d, n, c, vp(Wd), vp(ld), C.byref(r))
        assert col_rel_err(H.to_host(Wd, (d, d)), g[f"c{c}/W"]) < 1e-9
    gl = load_golden("losses")
    A = np.ascontiguousarray(gl["inv_sqrtm/A"])
    Ad, od = H.to_device(A), H.alloc(A.nbytes)
    call(H, "ccz_inv_sqrtm", vp(Ad), A.shape[0], 1e-5, vp(od))
    np.testing.assert_allclose(H.to_host(od, A.shape), gl["inv_sqrtm/out_eps1e-5"], atol=1e-9)
    call(H, "ccz_inv_sqrtm", vp(Ad), A.shape[0], 0.5, vp(od))
    np.testing.assert_allclose(H.to_host(od, A.shape), gl["inv_sqrtm/out_eps0.5"], atol=1e-10)


def test_seam_functions(H):
    from conftest import col_rel_err, load_golden
    from cca_zoo_amd._utils import gevp, svd_whiten

    g = load_golden("linalg_seams")
    for c in (0.0, 0.25):
        xw, W = svd_whiten(g["X"], c)
        assert col_rel_err(W, g[f"c{c}/W"]) < 1e-9
        S = np.sign(np.sum(W * g[f"c{c}/W"], axis=0))
        np.testing.assert_allclose((xw * S)[:5], g[f"c{c}/X_white_head"], atol=1e-9)
        np.testing.assert_allclose(np.cov(xw, rowvar=False) if c == 0 else np.eye(1), np.eye(xw.shape[1]) if c == 0 else np.eye(1), atol=1e-9)
    w, V = gevp(g["A"], None, 5)
    np.testing.assert_allclose(w, g["gevp_std/w"], rtol=1e-10)
    assert col_rel_err(V, g["gevp_std/V"]) < 1e-8
    w, V = gevp(g["A"], g["B"], 5)
    np.testing.assert_allclose(w, g["gevp_gen/w"], rtol=1e-10)
    assert col_rel_err(V, g["gevp_gen/V"]) < 1e-8


@pytest.mark.parametrize("n,d,k,ld_extra,ldo_extra", [(9000, 272, 64, 0, 0), (8200, 256, 7, 8, 1), (20000, 1024, 32, 0, 0),
                                                       (8192, 512, 1, 4, 0), (300, 40, 5, 0, 0),
                                                       (12345, 2048, 64, 0, 0), (8193, 320, 17, 4, 3), (70001, 4096, 48, 0, 0)])
def test_transform_f32_projection_kernel(H, n, d, k, ld_extra, ldo_extra):
    """(X - mean) W for fp32 samples: the projection kernels (n >= 8192, k <= 64: the wave-private FIFO kernel for
    d % 32 == 0, the 256 x 64 staged tile for d % 16 == 0), ragged row tail, padded output columns, strided input /
    output -- and the generic path for the small case."""
    from cca_zoo_amd import _backend

    rng = np.random.default_rng(n + d + k)
    ld, ldo = d + ld_extra, k + ldo_extra
    Xp = (rng.standard_normal((n, ld)) + 0.3).astype(np.float32)
    mean = rng.standard_normal(d)
    W = rng.standard_normal((d, k))
    Xd, md, Wd = H.to_device(Xp), H.to_device(mean), H.to_device(W)
    out0 = np.full((n, ldo), 7.0, dtype=np.float32)
    od = H.to_device(out0)
    call(H, "ccz_transform", _backend.F32, vp(Xd), n, d, ld, vp(md), vp(Wd), k, vp(od), ldo)
    got = H.to_host(od, (n, ldo), dtype=np.float32)
    ref = (Xp[:, :d].astype(np.float64) - mean) @ W
    scale = np.abs(ref).max()
    assert np.abs(got[:, :k] - ref).max() < 2e-5 * scale * np.sqrt(d)
    if ldo_extra:
        assert np.all(got[:, k:] == 7.0)                  # padding columns untouched


@pytest.mark.parametrize("impl,nj1", [("1", "1"), ("2", "1"), ("3", "0")])
@pytest.mark.parametrize("n,d,k,ld_extra", [(8321, 512, 64, 0), (16500, 320, 20, 4)])
def test_transform_f32_projection_kernel_forms(H, monkeypatch, impl, nj1, n, d, k, ld_extra):
    """The A/B forms of the projection kernel behind ``CCZ_TALL_IMPL`` (1: a row per lane, rounds 2-5; 2: whole-line loads with
    256 rows per workgroup; 3, the default: 128 rows) and ``CCZ_TALL_NJ1=0`` (two column tiles also for k <= 32) give the
    default form's result to fp32 rounding (same products, same order of accumulation per output)."""
    from cca_zoo_amd import _backend

    rng = np.random.default_rng(n + d + k)
    ld = d + ld_extra
    Xp = (rng.standard_normal((n, ld)) + 0.3).astype(np.float32)
    mean, W = rng.standard_normal(d), rng.standard_normal((d, k))
    Xd, md, Wd = H.to_device(Xp), H.to_device(mean), H.to_device(W)
    outs = []
    for env in (None, (impl, nj1)):
        if env is not None:
            monkeypatch.setenv("CCZ_TALL_IMPL", env[0])
            monkeypatch.setenv("CCZ_TALL_NJ1", env[1])
        od = H.to_device(np.zeros((n, k), dtype=np.float32))
        call(H, "ccz_transform", _backend.F32, vp(Xd), n, d, ld, vp(md), vp(Wd), k, vp(od), k)
        outs.append(H.to_host(od, (n, k), dtype=np.float32))
    ref = (Xp[:, :d].astype(np.float64) - mean) @ W
    scale = np.abs(ref).max()
    assert np.abs(outs[1] - ref).max() < 2e-5 * scale * np.sqrt(d)
    assert np.abs(outs[1] - outs[0]).max() < 1e-5 * scale


@pytest.mark.parametrize("n,d,k,impl", [(65536, 288, 256, "1"), (65600, 512, 512, "1"), (65536, 256, 256, "0")])
def test_transform_f32_wide_output_kernels(H, n, d, k, impl, monkeypatch):
    """(X - mean) W with k a multiple of 256 and >= 256 output tiles: the 256 x 256-tile kernels (wave-private
    LDS-DMA FIFO by default, register-staged tile with CCZ_GEMM_NN_IMPL=0; the choice is read once per process, so
    the second variant is only exercised when this test runs first in a fresh process)."""
    from cca_zoo_amd import _backend

    monkeypatch.setenv("CCZ_GEMM_NN_IMPL", impl)
    rng = np.random.default_rng(n + d + k)
    Xp = (rng.standard_normal((n, d)) + 0.3).astype(np.float32)
    mean = rng.standard_normal(d)
    W = rng.standard_normal((d, k))
    Xd, md, Wd = H.to_device(Xp), H.to_device(mean), H.to_device(W)
    od = H.alloc(n * k * 4)
    call(H, "ccz_transform", _backend.F32, vp(Xd), n, d, d, vp(md), vp(Wd), k, vp(od), k)
    got = H.to_host(od, (n, k), dtype=np.float32)
    rows = np.r_[0:300, n // 2:n // 2 + 300, n - 300:n]               # first / middle / ragged last tile
    ref = (Xp[rows].astype(np.float64) - mean) @ W
    scale = np.abs(ref).max()
    assert np.abs(got[rows] - ref).max() < 2e-5 * scale * np.sqrt(d)
    assert np.isfinite(got).all()
