"""GPU parity of the blocked Jacobi EVD / SVD (csrc/evd_block.hip) at the sizes the seams name.

Reference seams: ``svd_whiten`` (cca_zoo/_utils/_linalg.py:28-40), ``_inv_sqrtm`` (cca_zoo/deep/objectives.py:9-21,
deep/_base.py:176-188), ``_BatchWhiten`` (cca_zoo/deep/_dcca_noi.py:62-67), ``np.linalg.svd(cross_cov)``
(cca_zoo/linear/_rcca.py:97).  Comparator: ``np.linalg.eigh`` / ``np.linalg.svd`` (LAPACK, the arithmetic the reference
itself calls) and the oracle's reference-form whitener.  Bars: eigenvalues 1e-10 * ||A||, residual
||A V - V L|| / ||A|| < 1e-12, ||V'V - I|| < 1e-12.
"""

import ctypes as C
import time

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


def _matrix(kind, d, seed):
    rng = np.random.default_rng(seed)
    if kind == "sym":                                    # indefinite, Wigner spectrum
        A = rng.standard_normal((d, d))
        return A + A.T
    if kind == "cov":                                    # a sample covariance with a graded spectrum
        X = rng.standard_normal((3 * d, d)) * np.linspace(2.0, 0.05, d)
        X -= X.mean(0)
        return X.T @ X / (3 * d - 1)
    if kind == "lowrank":                                # rank d / 3: a large exactly-degenerate cluster at 0
        X = rng.standard_normal((d // 3, d))
        return X.T @ X
    if kind == "pm":                                     # +/- eigenvalue pairs (MCCA with two views)
        T = rng.standard_normal((d // 2, d - d // 2))
        return np.block([[np.zeros((d // 2, d // 2)), T], [T.T, np.zeros((d - d // 2, d - d // 2))]])
    raise ValueError(kind)


def _syevj(H, A):
    d = A.shape[0]
    Ad, wd, Vd = H.to_device(A), H.alloc(d * 8), H.alloc(d * d * 8)
    sw = C.c_int(0)
    H.sync()
    t0 = time.perf_counter()
    call(H, "ccz_syevj", vp(Ad), d, vp(wd), vp(Vd), C.byref(sw))
    H.sync()
    ms = (time.perf_counter() - t0) * 1e3
    return H.to_host(wd, (d,)), H.to_host(Vd, (d, d)), sw.value, ms


def _check_evd(A, w, V, tag):
    d = A.shape[0]
    wr = np.linalg.eigvalsh(A)[::-1]
    nrm = np.abs(wr).max()                               # = ||A||_2
    ev = np.abs(w - wr).max() / nrm
    # Frobenius norms (they bound the 2-norms from above and cost no SVD of a 4096 x 4096 matrix)
    res = np.linalg.norm(A @ V.T - V.T * w) / np.linalg.norm(A)
    orth = np.linalg.norm(V @ V.T - np.eye(d))
    print(f"[evd] {tag}: eig {ev:.2e} resid {res:.2e} orth {orth:.2e}")
    assert np.all(np.diff(w) <= 0.0), "eigenvalues must come back in descending order"
    assert ev < 1e-10, (tag, ev)
    assert res < 1e-12, (tag, res)
    assert orth < 1e-12, (tag, orth)


@pytest.mark.parametrize("kind,d", [("sym", 161), ("sym", 200), ("cov", 256), ("pm", 300), ("lowrank", 384), ("sym", 512),
                                     ("cov", 512), ("cov", 1000), ("sym", 1024), ("cov", 2048)])
def test_syevj_block_sizes(H, kind, d):
    A = _matrix(kind, d, d)
    w, V, sw, ms = _syevj(H, A)
    print(f"[evd] syevj {kind} d={d}: {sw} sweeps, {ms:.2f} ms (first call: includes graph capture)")
    _check_evd(A, w, V, f"{kind} {d}")
    assert 1 <= sw <= 40


def test_syevj_4096(H):
    A = _matrix("cov", 4096, 7)
    w, V, sw, ms = _syevj(H, A)
    w2, V2, sw2, ms2 = _syevj(H, A)
    print(f"[evd] syevj cov d=4096: {sw} sweeps, {ms:.1f} ms first / {ms2:.1f} ms second call")
    _check_evd(A, w, V, "cov 4096")
    np.testing.assert_array_equal(w, w2)                 # deterministic: no atomics in the arithmetic


def test_syevj_block_timings(H):
    """Warm timings (second call: graphs replayed) of the sizes VERDICT r3 item 1 names; printed, and bounded loosely
    so that a regression to the launch-per-round path (seconds) fails."""
    out = {}
    for d in (512, 1024, 2048):
        A = _matrix("cov", d, 11)
        _syevj(H, A)
        ts = [_syevj(H, A)[3] for _ in range(3)]
        out[d] = min(ts)
    print("[evd] warm syevj ms:", {k: round(v, 2) for k, v in out.items()})
    assert out[512] < 50 and out[1024] < 150 and out[2048] < 600


def test_syevj_block_edge_cases(H):
    d = 320
    w, V, sw, _ = _syevj(H, np.zeros((d, d)))
    assert np.all(w == 0.0) and np.allclose(V, np.eye(d)) and sw == 1
    w, V, sw, _ = _syevj(H, np.diag(np.arange(d, 0, -1.0)))
    np.testing.assert_array_equal(w, np.arange(d, 0, -1.0))
    assert sw == 1
    A = np.eye(d)
    A[5, 300] = A[300, 5] = np.nan
    with pytest.raises(ValueError, match="non-finite"):
        _syevj(H, A)
    # the handle is usable afterwards
    A = _matrix("sym", d, 3)
    w, V, _, _ = _syevj(H, A)
    _check_evd(A, w, V, "after the error")


@pytest.mark.parametrize("d", [512, 1024, 2048])
def test_whitener_and_inv_sqrtm_at_seam_sizes(H, d):
    """ccz_whitener (svd_whiten's W) and ccz_inv_sqrtm (_inv_sqrtm / _BatchWhiten) against eigh in float64."""
    rng = np.random.default_rng(d)
    n = 4 * d
    X = rng.standard_normal((n, d)) * np.linspace(1.5, 0.2, d)
    X -= X.mean(0)
    G = X.T @ X
    Gd = H.to_device(np.ascontiguousarray(G))
    for c in (0.0, 0.1):
        Wd, ld, r = H.alloc(d * d * 8), H.alloc(d * 8), C.c_int64(0)
        call(H, "ccz_whitener", vp(Gd), d, n, c, vp(Wd), vp(ld), C.byref(r))
        W, lam = H.to_host(Wd, (d, d)), H.to_host(ld, (d,))
        lr, Vr = np.linalg.eigh(G / (n - 1))
        lr, Vr = lr[::-1], Vr[:, ::-1]
        np.testing.assert_allclose(lam, lr, atol=1e-10 * lr[0])
        # W' ((1 - c) C + c I) W = I and W = V diag(...) up to signs: compare the whitening property and per-column directions
        R = (1 - c) * G / (n - 1) + c * np.eye(d)
        np.testing.assert_allclose(W.T @ R @ W, np.eye(d), atol=1e-9)
        Wr = Vr / np.sqrt((1 - c) * lr + c)
        S = np.sign(np.sum(W * Wr, axis=0))
        err = np.linalg.norm(W * S - Wr, axis=0) / np.linalg.norm(Wr, axis=0)
        assert np.median(err) < 1e-9 and err.max() < 1e-6, (c, err.max())
    A = G / (n - 1)
    Ad, od = H.to_device(np.ascontiguousarray(A)), H.alloc(A.nbytes)
    for eps in (1e-5, 0.5):
        call(H, "ccz_inv_sqrtm", vp(Ad), d, eps, vp(od))
        out = H.to_host(od, (d, d))
        lr, Vr = np.linalg.eigh(A)
        ref = (Vr / np.sqrt(np.maximum(lr, eps))) @ Vr.T
        np.testing.assert_allclose(out, ref, atol=1e-9 * np.abs(ref).max())


@pytest.mark.parametrize("d", [96, 100, 128, 160])
def test_whitener_between_the_one_workgroup_kernels_and_the_block_sizes(H, d):
    """PSD seams at 96 <= d <= 160: too wide for the one-workgroup ONE-sided kernel, and the two-sided one-workgroup kernel is
    not used for PSD inputs -- they take the blocked Jacobi with only 4 or 6 blocks (one or three pairs per round)."""
    rng = np.random.default_rng(d)
    n = 3 * d
    X = rng.standard_normal((n, d)) * np.linspace(1.5, 0.3, d)
    X -= X.mean(0)
    G = np.ascontiguousarray(X.T @ X)
    Gd = H.to_device(G)
    Wd, ld, r = H.alloc(d * d * 8), H.alloc(d * 8), C.c_int64(0)
    call(H, "ccz_whitener", vp(Gd), d, n, 0.1, vp(Wd), vp(ld), C.byref(r))
    W, lam = H.to_host(Wd, (d, d)), H.to_host(ld, (d,))
    lr = np.linalg.eigvalsh(G / (n - 1))[::-1]
    np.testing.assert_allclose(lam, lr, atol=1e-12 * lr[0])
    R = 0.9 * G / (n - 1) + 0.1 * np.eye(d)
    np.testing.assert_allclose(W.T @ R @ W, np.eye(d), atol=1e-11)


def test_svd_whiten_on_a_config2_view(H):
    """svd_whiten on a 1e5 x 1024 fp32 view (configs[1]) against the oracle's reference-form thin-SVD whitener."""
    from cca_zoo_amd._utils import svd_whiten
    from oracle import reference_form as rf

    rng = np.random.default_rng(5)
    n, d = 100_000, 1024
    X = (rng.standard_normal((n, 64)) @ rng.standard_normal((64, d)) + rng.standard_normal((n, d))).astype(np.float32)
    X -= X.mean(0, dtype=np.float64).astype(np.float32)
    xw, W = svd_whiten(X, 0.1)
    xw_ref, W_ref = rf.thin_svd_whitener(X.astype(np.float64), 0.1)
    S = np.sign(np.sum(W * W_ref, axis=0))
    err = np.linalg.norm(W * S - W_ref, axis=0) / np.linalg.norm(W_ref, axis=0)
    W64 = W.astype(np.float64)
    inv_err = np.linalg.norm(W64 @ W64.T - W_ref @ W_ref.T) / np.linalg.norm(W_ref @ W_ref.T)
    print(f"[evd] svd_whiten 1e5 x 1024 fp32: signal columns {err[:64].max():.2e}, all columns median {np.median(err):.2e}; "
          f"W W' = R^-1 {inv_err:.2e}")
    # float32 view, bar 1e-3 (north star, float32, sign-aligned).  The 64 signal directions are separated and compare per
    # column; the 960 noise directions have eigenvalue spacings ~ 4e-4 of the spectrum, where float32 data moves individual
    # eigenvectors by more than the bar in ANY solver -- they are compared through the rotation-invariant W W' = R^-1.
    assert err[:64].max() < 1e-4
    assert inv_err < 1e-3
    R = 0.9 * (X.astype(np.float64).T @ X.astype(np.float64)) / (n - 1) + 0.1 * np.eye(d)
    np.testing.assert_allclose(W64.T @ R @ W64, np.eye(d), atol=2e-3)
    xw_err = np.linalg.norm(xw.astype(np.float64)[:, :64] * S[:64] - xw_ref[:, :64]) / np.linalg.norm(xw_ref[:, :64])
    assert xw_err < 1e-3, xw_err


@pytest.mark.parametrize("shape", [(300, 40), (512, 512), (1024, 1024), (4096, 1024), (1000, 1500)])
def test_gesvj_block_sizes(H, shape):
    rng = np.random.default_rng(shape[0] + shape[1])
    A = rng.standard_normal(shape)
    p, q = shape
    r = min(p, q)
    Ad, Ud, sd, Vd = H.to_device(A), H.alloc(p * r * 8), H.alloc(r * 8), H.alloc(r * q * 8)
    sw = C.c_int(0)
    H.sync()
    t0 = time.perf_counter()
    call(H, "ccz_gesvj", vp(Ad), p, q, vp(Ud), vp(sd), vp(Vd), C.byref(sw))
    H.sync()
    ms = (time.perf_counter() - t0) * 1e3
    U, s, Vt = H.to_host(Ud, (p, r)), H.to_host(sd, (r,)), H.to_host(Vd, (r, q))
    sr = np.linalg.svd(A, compute_uv=False)
    print(f"[evd] gesvj {p} x {q}: {sw.value} sweeps, {ms:.1f} ms; sigma err {np.abs(s - sr).max() / sr[0]:.2e}")
    np.testing.assert_allclose(s, sr, atol=1e-10 * sr[0])
    assert np.linalg.norm(U * s @ Vt - A) < 1e-11 * np.linalg.norm(A) * 10
    assert np.linalg.norm(U.T @ U - np.eye(r)) < 1e-10
    assert np.linalg.norm(Vt @ Vt.T - np.eye(r)) < 1e-10


def test_gesvj_block_rank_deficient_and_graded(H):
    """One-sided block Jacobi on inputs the cross-covariance seam can produce: rank 100 of 512 (a 412-fold singular value 0: rows
    that have sunk below 1e-14 of the largest must rest, not rotate noise) and a graded spectrum over 12 decades (relative accuracy of
    the small singular values is what one-sided Jacobi is for)."""
    rng = np.random.default_rng(3)
    for tag in ("rank_deficient", "graded"):
        if tag == "rank_deficient":
            A = rng.standard_normal((512, 100)) @ rng.standard_normal((100, 512))
        else:
            U0, _ = np.linalg.qr(rng.standard_normal((384, 384)))
            V0, _ = np.linalg.qr(rng.standard_normal((384, 384)))
            A = (U0 * np.logspace(0, -12, 384)) @ V0.T
        p, q = A.shape
        r = min(p, q)
        Ad, Ud, sd, Vd = H.to_device(np.ascontiguousarray(A)), H.alloc(p * r * 8), H.alloc(r * 8), H.alloc(r * q * 8)
        sw = C.c_int(0)
        call(H, "ccz_gesvj", vp(Ad), p, q, vp(Ud), vp(sd), vp(Vd), C.byref(sw))
        U, s, Vt = H.to_host(Ud, (p, r)), H.to_host(sd, (r,)), H.to_host(Vd, (r, q))
        sr = np.linalg.svd(A, compute_uv=False)
        print(f"[evd] gesvj {tag}: {sw.value} sweeps; sigma err {np.abs(s - sr).max() / sr[0]:.2e}")
        np.testing.assert_allclose(s, sr, atol=1e-11 * sr[0])
        assert np.linalg.norm(U * s @ Vt - A) < 1e-10 * np.linalg.norm(A)
        # (directions of singular value 0 come back as zero rows of the long factor -- ccz_gesvj's contract since round 1:
        # the rCCA seam only consumes the leading ones -- so orthonormality is asserted on the numerically non-zero part)
        nz = int((sr > 1e-10 * sr[0]).sum())
        assert nz == (100 if tag == "rank_deficient" else int((sr > 1e-10).sum()))
        assert np.linalg.norm(Vt[:nz] @ Vt[:nz].T - np.eye(nz)) < 1e-9
        assert np.linalg.norm(U[:, :nz].T @ U[:, :nz] - np.eye(nz)) < 1e-9
        if tag == "graded":
            lead = sr > 1e-9                                        # relative accuracy down to where LAPACK itself is accurate
            assert np.abs(s[lead] / sr[lead] - 1.0).max() < 1e-6
        # measured: 27 sweeps (rank 100 of 512), 44 sweeps (12 decades, no preconditioning / row sorting yet); the solver's limit is 60
        assert sw.value <= 60
