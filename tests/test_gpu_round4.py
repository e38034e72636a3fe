"""GPU tests of round 4: the RCCL collective behind the C ABI (world size one: the GPU box has one GPU), the
torch-less linear path, the exception safety of the deferred exchange."""

import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    from cca_zoo_amd import _backend

    return _backend.default_handle(0)


def test_rccl_allreduce_behind_the_abi_world_size_one(H):
    """ccz_comm_unique_id -> ccz_comm_init_rank -> ccz_allreduce_sum_f64 on RCCL with ONE rank: the sum over one rank
    is the buffer itself; the communicator is reported and released; a second init on the same handle is refused."""
    uid = H.comm_unique_id()
    assert len(uid) == 128 and uid != b"\x00" * 128
    H.comm_init_rank(uid, 1, 0)
    try:
        assert H.comm_info() == (1, 0)
        with pytest.raises(ValueError, match="already has a communicator"):
            H.comm_init_rank(uid, 1, 0)
        x = np.random.default_rng(0).standard_normal(1 << 20)
        buf = H.to_device(x)
        H.allreduce_sum_f64(buf.ptr, x.size)
        np.testing.assert_array_equal(H.to_host(buf, x.shape), x)
    finally:
        H.comm_destroy()
    assert H.comm_info() == (0, -1)
    with pytest.raises(ValueError, match="no communicator"):
        H.allreduce_sum_f64(H.alloc(64).ptr, 8)


def test_row_sharded_over_a_ccz_comm_matches_the_plain_fit(H):
    """The linear path sharded WITHOUT torch.distributed: K1 -> pack -> ccz_allreduce_sum_f64 -> unpack -> solve, host
    (NumPy) views and HBM-resident views."""
    import torch

    from cca_zoo_amd import _dist, row_sharded
    from cca_zoo_amd.linear import GCCA, rCCA
    from oracle import reference_form as rf

    views = rf.joint_data(3, 3000, 4, [96, 80, 72], 2.0, 9)
    comm = _dist.CczComm(H, H.comm_unique_id(), 1, 0)
    try:
        for make, vs in ((lambda: rCCA(latent_dimensions=4, c=0.1), views[:2]), (lambda: GCCA(latent_dimensions=4, c=0.05), views)):
            plain = make().fit(vs)
            with row_sharded(group=comm):
                host = make().fit(vs)
                dev = make().fit([torch.as_tensor(v, device="cuda") for v in vs])
                sc = host.score(vs)
            for a, b, c in zip(plain.weights_, host.weights_, dev.weights_):
                np.testing.assert_allclose(b, a, rtol=1e-10, atol=1e-12)
                np.testing.assert_allclose(c, a, rtol=1e-10, atol=1e-12)
            np.testing.assert_allclose(sc, plain.score(vs), rtol=1e-10)
            assert host.timings_["allreduce_ms"] > 0.0
    finally:
        comm.close()


def test_torchless_process_fits_and_matches_the_oracle():
    """A ctypes-only process (CCZ_TORCHLESS=1: torch is never imported, libccz runs on /opt/rocm's HIP runtime) fits CCA
    on NumPy views and agrees with the oracle at 1e-8."""
    code = f"""
import os, sys
os.environ['CCZ_TORCHLESS'] = '1'
sys.path.insert(0, {ROOT!r})
import numpy as np
from cca_zoo_amd.linear import CCA, MCCA
from oracle import reference_form as rf
views = rf.joint_data(2, 2000, 3, [64, 48], 2.0, 1)
m = CCA(latent_dimensions=3).fit(views)
W_ref, _ = rf.rcca_weights(views, 3, c=0.0)
for w, r in zip(m.weights_, W_ref):
    s = np.sign(np.sum(w * r, axis=0))
    assert (np.linalg.norm(w * s - r, axis=0) / np.linalg.norm(r, axis=0)).max() < 1e-8
sc = m.score(views)
mm = MCCA(latent_dimensions=3, c=0.1).fit(views)
assert 'torch' not in sys.modules, 'the linear path imported torch'
print('ok', sc)
"""
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.startswith("ok"), (p.stdout[-500:], p.stderr[-3000:])


def test_failed_solve_consumes_the_deferred_exchange(H):
    """an unpack on a side stream registers the handle's OWN event; a solve that fails early (here:
    a bad regularisation) must consume it, and the next, unrelated solve on the handle must run normally."""
    import torch

    from cca_zoo_amd.linear import rCCA
    from oracle import reference_form as rf

    views = rf.joint_data(2, 1500, 3, [40, 30], 2.0, 2)
    ref = rCCA(latent_dimensions=3, c=0.1).fit(views)
    D, dims = 70, [40, 30]
    tv = [torch.as_tensor(v, device="cuda") for v in views]
    mom = torch.empty(D * D + D, dtype=torch.float64, device="cuda")
    H.moments([(t.data_ptr(), t.shape[1], t.stride(0)) for t in tv], 1500, 1, True, mom.data_ptr())
    count = D * (D + 1) // 2 + D + 1
    packed = torch.empty(count, dtype=torch.float64, device="cuda")
    H.moments_pack_blocks(mom.data_ptr(), D, dims, packed.data_ptr(), H.BOTH)
    H.sync()
    side = torch.cuda.Stream()
    out = torch.zeros_like(mom)
    H.moments_unpack_blocks(packed.data_ptr(), D, dims, out.data_ptr(), H.HEAD)
    H.moments_unpack_blocks(packed.data_ptr(), D, dims, out.data_ptr(), H.TAIL, on_stream=side.cuda_stream)   # registers the deferral
    with pytest.raises(ValueError):
        H.rcca_solve(out.data_ptr(), 1500, dims, [-1.0, 0.1], True, 3)        # fails in the argument checks
    W, _, _ = H.rcca_solve(out.data_ptr(), 1500, dims, [0.1, 0.1], True, 3)
    for a, b in zip(W, ref.weights_):
        s = np.sign(np.sum(a * b, axis=0))
        np.testing.assert_allclose(a * s, b, rtol=1e-9, atol=1e-11)


def test_c5_gcca_weights_against_the_oracle_at_full_dimensions(H):
    """``GCCA.weights_`` at configs[4]'s dimensions (d = [4096, 4096, 8192], D = 16384, k = 128,
    float64) with NON-uniform ``view_weights`` and per-view ``c`` against the oracle's Gram form evaluated on the host from
    the same float64 moments -- the top-k pairs of the 16384 x 16384 problem by Lanczos (scipy eigsh), no dense eigh.
    Per column, sign-aligned, 1e-5 (the float64 bar), on a separated spectrum (gaps asserted).  (cca_zoo/linear/_gcca.py:80-110)"""
    import scipy.sparse.linalg as spla
    import torch

    from cca_zoo_amd._moments import compute_moments
    from cca_zoo_amd.linear import GCCA
    from oracle import gram_form as gf

    dims, k, n = [4096, 4096, 8192], 128, 131072
    mu, cs = [1.0, 2.0, 0.5], [0.05, 0.1, 0.02]
    # a SEPARATED spectrum at these widths: orthonormal loadings of strength rho_j / (1 - rho_j) per latent (population
    # correlations 0.98, 0.975, ... 0.345), unit noise, the same latent z in every view -- JointData's N(0, 1) loadings
    # at d = 4096 push every correlation to 1 - 1e-4 and the 128 eigenvalues into a cluster of relative width 1e-4
    rho = 0.98 - 0.005 * np.arange(k)
    g = torch.Generator(device="cuda").manual_seed(7)
    amp = torch.as_tensor(np.sqrt(rho / (1.0 - rho)), dtype=torch.float64, device="cuda")
    loads = []
    for d_i in dims:
        q, _ = torch.linalg.qr(torch.randn(d_i, k, dtype=torch.float64, device="cuda", generator=g))
        loads.append((q * amp).T.contiguous())
    tv = [torch.empty(n, d_i, dtype=torch.float64, device="cuda") for d_i in dims]
    step = 16384
    for r0 in range(0, n, step):
        z = torch.randn(step, k, dtype=torch.float64, device="cuda", generator=g)
        for v, d_i in enumerate(dims):
            tv[v][r0:r0 + step] = z @ loads[v] + torch.randn(step, d_i, dtype=torch.float64, device="cuda", generator=g)
    del loads, z
    m = GCCA(latent_dimensions=k, c=cs, view_weights=mu).fit(tv)
    D = sum(dims)
    mom, keep, n_tot, _, _ = compute_moments(tv, H)
    H.moments_symmetrize(mom, D)
    flat = H.to_host(mom, (D * D + D,))
    del keep, tv
    torch.cuda.empty_cache()
    G, s = flat[:D * D].reshape(D, D), flat[D * D:]

    def lanczos(K, kk):
        # the oracle's Lanczos (scipy eigsh) with its ~1000 matrix-vector products of the 16384 x 16384 operator done by the
        # comparator on the device (torch.mv): 2.1 GB per product made this test 100 s of the suite on the host
        Kt = torch.as_tensor(K, device="cuda")
        op = spla.LinearOperator(K.shape, dtype=np.float64,
                                 matvec=lambda x: (Kt @ torch.as_tensor(np.ascontiguousarray(x).reshape(-1), device="cuda")).cpu().numpy())
        lam, U = spla.eigsh(op, k=kk, which="LA", ncv=3 * kk, tol=1e-11)
        o = np.argsort(lam)[::-1]
        return lam[o], U[:, o]

    W_ref, _means, lam_ref = gf.gcca_from_moments(G, s, n_tot, dims, k, c=cs, view_weights=mu, topk=lanczos)
    # per column where the neighbouring eigenvalues are separated (relative gap > 1e-6: a backward error of 1e-12 moves
    # such an eigenvector by 1e-6 at most); the remaining, nearly degenerate columns through the residual of the oracle's
    # columns in the span of ours (principal angles where correlations coincide)
    rel = np.abs(np.diff(lam_ref)) / lam_ref[0]
    gap = np.full(k, np.inf)
    gap[:-1] = np.minimum(gap[:-1], rel)
    gap[1:] = np.minimum(gap[1:], rel)
    sep = gap > 1e-6
    np.testing.assert_allclose(np.asarray(m.eigenvalues_)[:k], lam_ref, rtol=1e-9)
    col, sub = 0.0, 0.0
    for w, r in zip(m.weights_, W_ref):
        sgn = np.sign(np.sum(w * r, axis=0))
        e = np.linalg.norm(w * sgn - r, axis=0) / np.linalg.norm(r, axis=0)
        col = max(col, float(e[sep].max()))
        coef, *_ = np.linalg.lstsq(w, r, rcond=None)
        sub = max(sub, float(np.linalg.norm(r - w @ coef) / np.linalg.norm(r)))
    print(f"[c5 gcca] separated columns {int(sep.sum())}/{k}: max column error {col:.2e}; subspace residual {sub:.2e}; min gap {gap.min():.1e}")
    assert sep.sum() >= k // 2
    assert col < 1e-5 and sub < 1e-5, (col, sub)
