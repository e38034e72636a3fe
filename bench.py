#!/usr/bin/env python3
"""Headline benchmark: CCA fit()/sec at n=1e6, 2 views x 4096 features, k=64 (BASELINE.json).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full ``CCA(latent_dimensions=64).fit(views)`` on synthetic latent-variable views already resident
in HBM (JointData model): column sums + K1 Gram on the MFMA pipe, (N > 1) one RCCL all-reduce of the
packed moments, the device solves, weights back on the host.  With N ranks the n rows of ONE data set are split
contiguously across the ranks (``sample_device(row0=...)``: the data do not depend on N; total work fixed ->
"scaling": "strong").  Rank 0 prints ONE JSON line.

Parity gate: before anything is printed the fitted model is checked on the very views that were timed
(``check_fit_properties``): rows regenerated on the host by the NumPy restatement of the generator, K1 against a
float64 Gram of those rows, and the size-independent properties of a CCA solution on all n rows.  A failed gate
aborts the run (no JSON line).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6, "bf16": 2500.0}   # dense MFMA peaks, MI355X_MICROARCH.md / datasheet
DATA_SEED = 20260


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", dest="n", type=int, default=1_000_000)
    ap.add_argument("--d", "--dim", dest="d", type=int, default=4096)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--k1-route", choices=["auto", "fp32", "bf16x2"], default=os.environ.get("CCZ_BENCH_K1_ROUTE", "auto"),
                    help="arithmetic route of fp32 views through K1 (include/ccz.h: ccz_k1_route); auto = split-bf16 where it pays")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra configurations (C2, C3, C5, f64, losses, grid search)")
    ap.add_argument("--no-dcca", action="store_true")
    ap.add_argument("--with-sharded-dcca", action="store_true",
                    help="N > 1 only: also time CCALoss fwd+bwd on the batch sharded over the ranks (extra collectives "
                         "after the timed fits; off by default so that nothing can delay the headline result)")
    ap.add_argument("--no-cpu-mcca", action="store_true",
                    help="skip the reference-structured MCCA / GCCA comparators (BASELINE configs[2] / [4] on bounded samples: ~1 min of CPU)")
    ap.add_argument("--cpu-mcca", action="store_true", help="(kept for compatibility: the MCCA comparator is on by default)")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of the rCCA CPU comparator's sample (0: 2 d)")
    ap.add_argument("--cpu-runs", type=int, default=3, help="repeats of the rCCA CPU comparator at n = 2 d (a run is ~25 s; the median is reported)")
    ap.add_argument("--no-gates", action="store_true", help="skip the parity gates of the extras (the headline gate always runs)")
    ap.add_argument("--only", default="", help="comma-separated subset of the extras to run (routes, transform, dcca, grid, host, metric_loss, configs, evd)")
    ap.add_argument("--transport", choices=["torch", "ccz", "gloo-staged"], default=os.environ.get("CCZ_BENCH_TRANSPORT", "torch"),
                    help="exchange step of the sharded fit: torch.distributed (nccl = RCCL) all-reduces, or libccz's own RCCL collective "
                         "behind the C ABI (ccz_moments_exchange); both run the two-part exchange that overlaps the factorization")
    ap.add_argument("--launch-test", action="store_true",
                    help="(CPU test of the launcher) rendezvous over gloo, one all-reduce, ONE JSON line on rank 0; no GPU work")
    return ap.parse_args()


def dist_all_reduce(t, op=None):
    """``dist.all_reduce`` that also serves ``--transport gloo-staged`` (two ranks on ONE GPU: CUDA tensors over a gloo group go
    through a host copy; cca_zoo_amd._dist.staged_over_gloo is the product-side twin)."""
    import torch.distributed as dist

    op = op if op is not None else dist.ReduceOp.SUM
    if t.is_cuda and dist.get_backend() == "gloo":
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)


# ---------------------------------------------------------------------------------------------------------------
# parity gate
# ---------------------------------------------------------------------------------------------------------------
def k1_spot_check(views, jd, seed, row0=0, n_check=2048):
    """Generator + K1 on the first ``n_check`` local rows: the rows regenerated on the host by the NumPy restatement
    (``oracle.rng``) equal the device rows, and the device moments of those rows agree with their float64 Gram."""
    import numpy as np

    from cca_zoo_amd import _backend, _dist
    from cca_zoo_amd._moments import compute_moments
    from oracle import rng as orng

    rep = {}
    ndt = np.float32 if views[0].element_size() == 4 else np.float64
    m_rows = min(n_check, int(views[0].shape[0]))
    host = orng.joint_data_rows(jd._weights, jd._snr_per_view, jd.latent_scales, seed=seed, row0=row0, rows=m_rows, dtype=ndt)
    # the device forms the signal z W' with an fp32 (fp64) GEMM, the restatement in float64 then rounds: a few ulp
    # of the LARGEST summands -- compare relative to the largest element
    rep["generator_rel_diff"] = max(float(np.abs(v[:m_rows].cpu().numpy().astype(np.float64) - h.astype(np.float64)).max()
                                          / np.abs(h.astype(np.float64)).max()) for v, h in zip(views, host))
    ok = rep["generator_rel_diff"] < (4e-6 if ndt == np.float32 else 1e-12)
    # K1 on the checked rows (never sharded: a local quantity)
    h = _backend.handle_for(views)
    with _dist.unsharded():
        mom, keep, _, dims, _ = compute_moments([v[:m_rows] for v in views], h)
    D = int(sum(dims))
    flat = h.to_host(mom, (D * D + D,))
    del keep
    X = np.hstack([v.astype(np.float64) for v in host])
    iu = np.triu_indices(D)
    Gref = X.T @ X
    scale = np.sqrt(np.outer(np.diag(Gref), np.diag(Gref)))[iu]
    rep["k1_rel_err"] = float((np.abs(flat[:D * D].reshape(D, D)[iu] - Gref[iu]) / scale).max())
    rep["ok"] = bool(ok and rep["k1_rel_err"] < (2e-5 if ndt == np.float32 else 1e-12))
    return rep


def k1_rates(h, flop, g_ms, kind):
    """Rates of the last K1 launch on handle ``h``: ``flop`` = ALGORITHMIC flops n D (D+1), ``g_ms`` = the whole K1 (for the
    split-bf16 route: split pass + MFMA kernel + reduce).  The fp32 / fp64 kernels execute the algorithmic flops on their own
    pipe; the split route executes THREE bf16 products per algorithmic one (hi'hi + hi'mid + mid'hi), so its roofline is
    3 x flop over the MFMA kernel's own time against the dense bf16 peak, with the algorithmic rate beside it."""
    route, split_ms, mfma_ms, reduce_ms = h.moments_last_route()
    alg = flop / (g_ms * 1e-3) / 1e12
    if route == "bf16x2" and mfma_ms > 0:
        ex = 3.0 * flop / (mfma_ms * 1e-3) / 1e12
        return {"k1_route": route, "gram_tflops": alg, "gram_executed_bf16_tflops": ex, "gram_frac_of_peak": ex / PEAK_TFLOPS["bf16"],
                "gram_peak_tflops": PEAK_TFLOPS["bf16"], "gram_stages_ms": {"split": split_ms, "mfma": mfma_ms, "reduce": reduce_ms}}
    return {"k1_route": route, "gram_tflops": alg, "gram_frac_of_peak": alg / PEAK_TFLOPS[kind], "gram_peak_tflops": PEAK_TFLOPS[kind]}


def k1_routes_extra(h, views, make_model, chunk=16384, fits=3):
    """Both arithmetic routes of K1 on ALL rows of the timed views: max relative error of a Gram entry against float64
    moments of the same fp32 rows (device float64 GEMM in row chunks: the vendor's, a comparator outside every timed
    region) and the fit time with the route forced.  The gate of the split route: its error is no larger than the fp32
    kernel's on the same rows -- else it would be narrower arithmetic than the reference's float32 path."""
    import numpy as np
    import torch

    n = int(views[0].shape[0])
    dims = [int(v.shape[1]) for v in views]
    D = sum(dims)
    dev = views[0].device
    G64 = torch.zeros(D, D, dtype=torch.float64, device=dev)
    for r0 in range(0, n, chunk):
        xc = torch.cat([v[r0:r0 + chunk] for v in views], dim=1).double()
        G64.addmm_(xc.T, xc)
        del xc
    dg = torch.diag(G64)
    iu = torch.triu(torch.ones(D, D, dtype=torch.bool, device=dev))
    prev = h.k1_route(None)
    out = {"rows": n, "D": D, "comparator": "float64 Gram of the same rows (torch.addmm on the device, row chunks of %d)" % chunk}
    mom = torch.empty(D * D + D, dtype=torch.float64, device=dev)
    try:
        for route in ("fp32", "bf16x2"):
            h.k1_route(route)
            torch.cuda.synchronize()
            h.moments([(v.data_ptr(), v.shape[1], v.stride(0)) for v in views], n, 0, True, mom.data_ptr())
            h.sync()
            g_ms = h.moments_last_ms()[0]
            G = mom[:D * D].view(D, D)
            err = torch.zeros((), dtype=torch.float64, device=dev)
            for i0 in range(0, D, 1024):                       # row blocks: no second and third D x D temporaries
                blk = (G[i0:i0 + 1024] - G64[i0:i0 + 1024]).abs() / torch.sqrt(dg[i0:i0 + 1024, None] * dg[None, :])
                err = torch.maximum(err, blk[iu[i0:i0 + 1024]].max())
            rec = {"k1_rel_err": float(err), "k1_ms": g_ms, **k1_rates(h, float(n) * D * (D + 1), g_ms, "f32")}
            ts = []
            for _ in range(fits + 1):
                t0 = time.perf_counter()
                make_model().fit(views)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            rec["fit_ms"] = float(np.median(ts[1:]))
            out[route] = rec
    finally:
        h.k1_route(prev)
    out["split_no_worse_than_fp32"] = bool(out["bf16x2"]["k1_rel_err"] <= out["fp32"]["k1_rel_err"])
    del G64, mom
    return out


def check_fit_properties(model, views, jd, seed, row0=0, sharded=False, n_check=2048):
    """Is ``model`` (a c = 0 CCA / rCCA fitted on ``views``) a CCA solution of THESE views?

    * generator + K1 (``k1_spot_check``) on the first ``n_check`` local rows;
    * solution (all rows; global under ``sharded``): every canonical variate has unit variance (w'C w = 1), the
      variates of a view are uncorrelated, the cross-view correlation matrix is diag(singular values), the training
      score equals the singular values, and those are non-increasing in (0, 1].
    Returns a dict with ``ok`` and the measured deviations."""
    import contextlib

    import numpy as np

    from cca_zoo_amd import _backend, row_sharded
    from cca_zoo_amd._moments import compute_moments

    rep = k1_spot_check(views, jd, seed, row0, n_check)
    ok = rep.pop("ok")
    tol = 1e-3 if views[0].element_size() == 4 else 1e-5
    h = _backend.handle_for(views)
    # the solution on all rows
    ctx = row_sharded() if sharded else contextlib.nullcontext()
    with ctx:
        zs = model.transform(views)
        k = int(zs[0].shape[1])
        momz, keepz, n_tot, _, _ = compute_moments(zs, h)
        K2 = len(zs) * k
        h.moments_symmetrize(momz, K2)
        fz = h.to_host(momz, (K2 * K2 + K2,))
        del keepz
        score = model.score(views)
    Gz, sz = fz[:K2 * K2].reshape(K2, K2), fz[K2 * K2:]
    Cz = (Gz - np.outer(sz, sz) / n_tot) / (n_tot - 1)
    sv = np.asarray(model.singular_values_, dtype=np.float64)
    rep["unit_variance_dev"] = float(np.abs(np.diag(Cz) - 1.0).max())
    within = max(float(np.abs(Cz[i * k:(i + 1) * k, i * k:(i + 1) * k] - np.eye(k)).max()) for i in range(len(zs)))
    rep["within_view_corr_dev"] = within
    rep["cross_view_dev"] = float(np.abs(Cz[:k, k:2 * k] - np.diag(sv)).max())
    rep["score_vs_singular_values"] = float(np.abs(np.asarray(score, dtype=np.float64) - sv).max())
    rep["singular_values"] = [float(sv[0]), float(sv[-1])]
    ok = ok and rep["unit_variance_dev"] < 3 * tol and within < 3 * tol and rep["cross_view_dev"] < 3 * tol
    ok = ok and rep["score_vs_singular_values"] < 3 * tol and bool(np.all(np.diff(sv) <= 1e-9)) and 0.0 < sv[-1] and sv[0] <= 1.0 + 1e-9
    # ---- THE top-k solution, not just A CCA solution: the pencil certificate of oracle/certificates.py
    # with the inertia count, on second moments of the timed views formed by a COMPARATOR (torch float64 products over
    # row chunks; summed over the ranks under sharding) -- not by the product's K1
    cert = topk_certificate(model, views, sharded)
    if cert is not None:                                   # (under sharding only rank 0 evaluates it; the verdict is shared below)
        rep.update(cert)
        ok = ok and cert["pencil_residual"] < tol and cert["pencil_orthonormality"] < tol
        ok = ok and cert["pencil_eigenvalues_above_lambda_k"] == k
    if sharded:
        # one verdict for all ranks: a rank that fails alone would leave the others waiting in the next collective
        import torch
        import torch.distributed as dist

        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=views[0].device)
        dist_all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item() > 0.5)
    rep["ok"] = bool(ok)
    return rep


def topk_certificate(model, views, sharded=False, chunk=32768):
    """Eigen-residual, B-orthonormality and the number of pencil eigenvalues above lambda_k (by inertia: one LDL'
    factorization on the host) for the two-view pencil of ``cca_zoo/linear/_rcca.py:92-100`` at c = 0, from float64
    moments that torch accumulates here.  Every rank takes part in the moments; rank 0 alone runs the factorization (returns ``None`` elsewhere)."""
    import numpy as np
    import torch

    from oracle import certificates as ct

    dims = [int(v.shape[1]) for v in views]
    D, n_loc, dev = sum(dims), int(views[0].shape[0]), views[0].device
    G = torch.zeros(D, D, dtype=torch.float64, device=dev)
    sm = torch.zeros(D, dtype=torch.float64, device=dev)
    for a in range(0, n_loc, chunk):
        X = torch.cat([v[a:a + chunk].to(torch.float64) for v in views], dim=1)
        G.addmm_(X.T, X)
        sm += X.sum(0)
        del X
    n_tot = torch.tensor([float(n_loc)], dtype=torch.float64, device=dev)
    if sharded:
        import torch.distributed as dist

        for t in (G, sm, n_tot):
            dist_all_reduce(t)
    if sharded:
        import torch.distributed as dist

        if dist.get_rank() != 0:                           # the host-side certificate is the same on every rank: rank 0 does it
            return None
    Gh, sh, n = G.cpu().numpy(), sm.cpu().numpy(), int(round(float(n_tot.item())))
    del G, sm
    torch.cuda.empty_cache()
    A, B = ct.rcca_pencil(Gh, sh, n, dims, [0.0] * len(dims))
    V = np.vstack(model.weights_).astype(np.float64) / np.sqrt(2.0)
    lam = np.asarray(model.singular_values_, dtype=np.float64)
    f32 = views[0].element_size() == 4
    limiter = None
    if sharded:                                            # the other ranks sit in an all-reduce: their cores are free
        try:
            from threadpoolctl import threadpool_limits

            affinity, quota = host_cores()
            limiter = threadpool_limits(limits=max(1, int(min(affinity, quota) if quota else affinity) - 2))
        except Exception:
            limiter = None
    try:
        r = ct.pencil_certificate(A, B, V, lam, delta=1e-4 if f32 else 1e-6, inertia=True)
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    return {"pencil_residual": float(r["residual"]), "pencil_orthonormality": float(r["orthonormality"]),
            "pencil_eigenvalues_above_lambda_k": int(r["n_above"]), "k": int(r["k"]),
            "moments_by": "torch float64 addmm over row chunks (comparator)"}


def solution_gate(model, views, est, c, jd=None, seed=None, inertia=True):
    """Parity gate of an extra configuration: (1) generator + K1 spot check on regenerated rows, (2) the fitted model is
    THE top-k solution of the symmetric-definite pencil the reference's estimator defines (``oracle.certificates``:
    eigen-residual, B-orthonormality, exactly k pencil eigenvalues above lambda_k by inertia) -- on the moments of the
    timed views themselves.  ``est``: "rcca" | "mcca" | "gcca"."""
    import numpy as np

    from cca_zoo_amd import _backend
    from cca_zoo_amd._moments import compute_moments
    from oracle import certificates as ct

    f32 = (views[0].element_size() if hasattr(views[0], "element_size") else views[0].itemsize) == 4   # torch tensor | ndarray
    tol = 1e-3 if f32 else 1e-7
    rep = {"tol": tol}
    ok = True
    if jd is not None:
        spot = k1_spot_check(views, jd, seed, 0, 2048 if sum(int(v.shape[1]) for v in views) <= 8192 else 1024)
        ok = spot.pop("ok")
        rep.update(spot)
    h = _backend.handle_for(views)
    mom, keep, n, dims, _ = compute_moments(views, h)
    D = int(sum(dims))
    h.moments_symmetrize(mom, D)
    flat = h.to_host(mom, (D * D + D,))
    del keep
    G, sm = flat[:D * D].reshape(D, D), flat[D * D:]
    cs = [float(c)] * len(dims)
    if est == "rcca":
        A, B = ct.rcca_pencil(G, sm, n, dims, cs)
        V, lam = np.vstack(model.weights_).astype(np.float64) / np.sqrt(2.0), np.asarray(model.singular_values_, dtype=np.float64)
    elif est == "mcca":
        A, B = ct.mcca_pencil(G, sm, n, dims, cs)
        V, lam = np.vstack(model.weights_).astype(np.float64) / np.sqrt(len(dims)), np.asarray(model.eigenvalues_, dtype=np.float64)
    else:
        lam = np.asarray(model.eigenvalues_, dtype=np.float64)
        A, B, V = ct.gcca_pencil(G, sm, n, dims, cs, model.weights_, lam)
    del flat, G
    r = ct.pencil_certificate(A, B, V, lam, delta=1e-4 if f32 else 1e-6, inertia=inertia)
    rep.update({"pencil_residual": r["residual"], "pencil_orthonormality": r["orthonormality"], "k": r["k"],
                "pencil_eigenvalues_above_lambda_k": r.get("n_above", "not counted")})
    ok = ok and r["residual"] < tol and r["orthonormality"] < tol and (not inertia or r["n_above"] == r["k"])
    ok = ok and bool(np.all(np.diff(lam) <= 1e-9 * max(1.0, abs(lam[0]))))
    rep["ok"] = bool(ok)
    return rep


# ---------------------------------------------------------------------------------------------------------------
# CPU comparators (reference-structured NumPy / torch paths, bounded samples)
# ---------------------------------------------------------------------------------------------------------------
def host_cores():
    """CPUs this process may actually use: scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    return n, quota


def cpu_baseline(n_full, d, k, sample_rows=0, runs=3):
    """The reference's rCCA structure (cca_zoo/linear/_rcca.py:92-100: thin SVD of each centred n x d view, whitened
    cross product, SVD of the d x d matrix T) from the oracle's own building blocks (``oracle.reference_form``), timed as
    a SLOPE: the n-dependent part (two thin SVDs + the n x d x d cross product) is measured at TWO
    sample sizes, n1 = 2 d rows (median of ``runs``, default 3) and n2 = 4 d rows (two runs), and the cost at the full n
    is  fixed + t(n1) + per_row * (n_full - n1)  with  per_row = (t(n2) - t(n1)) / (n2 - n1)  -- at n ~ d the "n-dependent"
    part still holds large d^3 terms of LAPACK's gesdd that do not grow with n, which a plain ratio n_full / n1 (round 3)
    scaled along with everything else.  The part that does not depend on n at all (SVD of T, back-multiplication) is
    timed on its own.  ~2.2 min of CPU on 16 cores."""
    import numpy as np

    from oracle import reference_form as rf

    affinity, quota = host_cores()
    cores = int(min(affinity, quota)) if quota else affinity
    # one BLAS thread per usable core: the default (one per LOGICAL cpu of the box, 256) oversubscribes a 16-cpu quota
    # so badly that the same SVD takes 5x longer
    try:
        from threadpoolctl import threadpool_limits

        limiter = threadpool_limits(limits=max(cores, 1))
    except Exception:
        limiter = None
    n1 = sample_rows if sample_rows > 0 else 2 * d
    n2 = 2 * n1
    views2 = [v.astype(np.float32) for v in rf.joint_data(2, n2, k, [d, d], 1.0, 0)]

    def data_part(rows):
        t0 = time.perf_counter()
        (X1, X2), _means = rf.center_views([v[:rows] for v in views2], True)
        X1w, W1 = rf.thin_svd_whitener(X1, 0.0)
        X2w, W2 = rf.thin_svd_whitener(X2, 0.0)
        T = X1w.T @ X2w / (X1.shape[0] - 1)
        return time.perf_counter() - t0, T, W1, W2, min(k, X1w.shape[1], X2w.shape[1])

    t_n1, t_n2, t_fixed = [], [], []
    try:
        for _ in range(max(1, runs)):
            td, T, W1, W2, kk = data_part(n1)
            t1 = time.perf_counter()
            U, _sv, Vt = np.linalg.svd(T, full_matrices=False)
            _W = [W1 @ U[:, :kk], W2 @ Vt[:kk].T]
            t_fixed.append(time.perf_counter() - t1)
            t_n1.append(td)
            del T, U, Vt, _W, W1, W2
        for _ in range(2 if runs > 1 else 1):
            t_n2.append(data_part(n2)[0])
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    td1, td2, tf = float(np.median(t_n1)), float(np.median(t_n2)), float(np.median(t_fixed))
    per_row = max((td2 - td1) / (n2 - n1), 0.0)
    full = tf + td1 + per_row * (n_full - n1)
    return {
        "value": 1.0 / full, "unit": "fit/s (extrapolated along the measured per-row slope)", "cores": max(cores, 1),
        "kind": "port",
        "sample": (f"the reference's rCCA structure (thin SVD per view, cross product, SVD of T: cca_zoo/linear/_rcca.py:92-100) "
                   f"from oracle.reference_form's building blocks on {n1} and {n2} rows (2 d, 4 d) of 2x{d} fp32 JointData, k={k}; "
                   f"{len(t_n1)} + {len(t_n2)} runs"),
        "n_dependent_s": {"rows": [n1, n2], "median_s": [td1, td2], "runs_s": [[round(x, 3) for x in t_n1], [round(x, 3) for x in t_n2]]},
        "per_row_s": per_row, "intercept_s": td1 - per_row * n1, "fixed_s": tf, "runs": len(t_n1),
        "measured_s": td1 + tf, "extrapolated_full_s": full,
        "extrapolation": f"fixed_s + t({n1}) + per_row_s * ({n_full} - {n1});  per_row_s = (t({n2}) - t({n1})) / {n2 - n1}",
        "ratio_extrapolation_s_for_comparison": tf + td1 * (n_full / n1),
        "blas_threads": max(cores, 1) if limiter is not None else None, "sched_affinity": affinity, "cgroup_cpu_quota": quota,
        "logical_cpus": os.cpu_count(),
    }


def cpu_c2_whole(k=32, c=0.1, n=100_000, d=1024):
    """BASELINE configs[1] WHOLE on the host, no extrapolation: the oracle's reference-form
    ``rcca_weights`` (cca_zoo/linear/_rcca.py:69-101) on the very views the GPU fits -- n = 1e5, 2 x 1024, float32 --
    timed, and the two solutions compared by subspace and score.  JointData's 32 correlations lie within 6e-6 of each
    other (no single column is defined at the float32 bar), so the per-column 1e-3 bar is exercised on a SECOND data set of
    the same shape whose correlations are separated: entry ``separated``."""
    import numpy as np
    import torch

    from cca_zoo_amd.datasets import JointData
    from cca_zoo_amd.linear import rCCA
    from oracle import reference_form as rf

    affinity, quota = host_cores()
    cores = int(min(affinity, quota)) if quota else affinity

    def compare(views, sep_factor):
        model = rCCA(latent_dimensions=k, c=c).fit(views)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model = rCCA(latent_dimensions=k, c=c).fit(views)
        torch.cuda.synchronize()
        gpu_s = time.perf_counter() - t0
        host = [v.cpu().numpy() for v in views]
        gpu_score = np.asarray(model.score(views), dtype=np.float64)
        try:
            from threadpoolctl import threadpool_limits

            limiter = threadpool_limits(limits=max(cores, 1))
        except Exception:
            limiter = None
        try:
            t0 = time.perf_counter()
            W_ref, means_ref = rf.rcca_weights(host, k, c=c)
            cpu_s = time.perf_counter() - t0
        finally:
            if limiter is not None:
                limiter.restore_original_limits()
        agree = weights_agreement(model.weights_, W_ref, model.singular_values_, 1e-3, sep_factor)
        ref_score = rf.mean_offdiag_corr([h.astype(np.float64) for h in host], W_ref, means_ref)
        return {"cpu_s": cpu_s, "cpu_fits_per_s": 1.0 / cpu_s, "gpu_fit_s": gpu_s, "gpu_over_cpu": cpu_s / gpu_s,
                "weights_vs_oracle": agree, "score_max_abs_diff": float(np.abs(gpu_score - np.asarray(ref_score)).max()),
                "singular_values": [float(model.singular_values_[0]), float(model.singular_values_[-1])],
                "agree_at_1e-3": bool(agree["ok"])}

    jd = JointData(n_views=2, n_samples=1, latent_dimensions=k, n_features=[d, d], random_state=1,
                   latent_scales=list(np.linspace(2.0, 0.5, k)))
    views = jd.sample_device(device="cuda", dtype=torch.float32, n_samples=n, seed=DATA_SEED + 1)
    out = {"config": f"configs[1] whole: rCCA n={n}, 2x{d}, k={k}, c={c}, float32", "cores": max(cores, 1), "extrapolated": False}
    out.update(compare(views, 100.0))
    del views
    torch.cuda.empty_cache()
    # the same shape with population correlations 0.97, 0.948, ... 0.30 (orthonormal loadings of strength rho / (1 - rho) per
    # latent, unit noise): sample gaps ~2e-2.  A column's first-order error is (moment error ~1e-6 .. 4e-6 for float32
    # views) / gap, so gaps above 8 x tol leave the 1e-3 bar well-posed per column (the same criterion as
    # tests/test_gpu_round3.py::test_ns_dimensions_per_column_on_a_separated_spectrum).
    rho = np.linspace(0.97, 0.30, k)
    g = torch.Generator(device="cuda").manual_seed(DATA_SEED + 5)
    amp = torch.as_tensor(np.sqrt(rho / (1.0 - rho)), dtype=torch.float64, device="cuda")
    loads = []
    for _ in range(2):
        q, _ = torch.linalg.qr(torch.randn(d, k, dtype=torch.float64, device="cuda", generator=g))
        loads.append((q * amp).T.contiguous())
    z = torch.randn(n, k, dtype=torch.float64, device="cuda", generator=g)
    sviews = [(z @ L + torch.randn(n, d, dtype=torch.float64, device="cuda", generator=g)).to(torch.float32) for L in loads]
    del z
    out["separated"] = compare(sviews, 8.0)
    out["separated"]["data"] = "orthonormal loadings, population correlations linspace(0.97, 0.30, 32), unit noise, float32"
    out["agree_at_1e-3"] = bool(out["agree_at_1e-3"] and out["separated"]["agree_at_1e-3"])
    return out


def cpu_mcca_baseline(n_full=1_000_000, d=2048, m=4, k=64, sample_rows=4096):
    """``oracle.reference_form.mcca_weights`` (PCA per view, np.cov of the projections, eps-floor, scipy eigh subset:
    cca_zoo/linear/_mcca.py:99-197) on ``sample_rows`` = 4 d rows of BASELINE configs[2]-shaped data.  The part that
    grows with n (the m thin SVDs and the covariance: O(n D^2)) is timed on its own so that the extrapolation to the
    full n only scales THAT part; the D^3 eigen-solve does not grow."""
    import numpy as np

    from oracle import reference_form as rf

    affinity, quota = host_cores()
    cores = int(min(affinity, quota)) if quota else affinity
    views = [v.astype(np.float32) for v in rf.joint_data(m, sample_rows, k, [d] * m, 1.0, 0)]
    try:
        from threadpoolctl import threadpool_limits

        limiter = threadpool_limits(limits=max(cores, 1))
    except Exception:
        limiter = None
    try:
        t0 = time.perf_counter()
        vs, _ = rf.center_views(views, True)
        fits = [rf._pca_full(v) for v in vs]
        rf._between_view_cov([f[2] for f in fits])
        t_data = time.perf_counter() - t0
        del fits, vs
        t0 = time.perf_counter()
        rf.mcca_weights(views, k, c=0.0)
        t_all = time.perf_counter() - t0
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    t_fixed = max(t_all - t_data, 0.0)
    full = t_fixed + t_data * (n_full / sample_rows)
    return {"metric": f"CPU MCCA fit (reference structure), {m} x {d}, k={k}", "sample_rows": sample_rows, "cores": max(cores, 1),
            "measured_s": t_all, "data_dependent_s": t_data, "eigen_solve_s": t_fixed,
            "extrapolated_full_s": full, "extrapolation": f"eigen_solve_s + data_dependent_s * {n_full}/{sample_rows}",
            "value": 1.0 / full, "unit": "fit/s (extrapolated)"}


def cpu_gcca_baseline(n_rows=2048, dims=(2048, 2048, 4096), k=128, n_full=2_000_000, full_dims=(4096, 4096, 8192)):
    """``oracle.reference_form.gcca_weights`` -- the reference's n x n formulation (cca_zoo/linear/_gcca.py:80-110:
    Q = sum_i X_i R_i^-1 X_i', top-k eigenvectors, pinv) -- at the largest size a ~20 s CPU budget allows: BASELINE
    configs[4]'s three views at HALF width and n = 2048 rows.  The formulation is O(n^2 d + n^3) in time and n^2 in
    memory: at the config's n = 2e6 the n x n matrix alone would be 32 TB, so there is no extrapolation -- the
    reference cannot run configs[4]; the measured sample is context only."""
    import numpy as np

    from oracle import reference_form as rf

    affinity, quota = host_cores()
    cores = int(min(affinity, quota)) if quota else affinity
    views = rf.joint_data(len(dims), n_rows, k, list(dims), 1.0, 0)
    try:
        from threadpoolctl import threadpool_limits

        limiter = threadpool_limits(limits=max(cores, 1))
    except Exception:
        limiter = None
    try:
        t0 = time.perf_counter()
        rf.gcca_weights(views, k, c=0.1)
        dt = time.perf_counter() - t0
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    return {"metric": f"CPU GCCA fit (reference n x n structure), n={n_rows}, d={list(dims)}, k={k}, float64", "cores": max(cores, 1),
            "measured_s": dt, "extrapolated_full_s": None,
            "note": (f"configs[4] itself (n={n_full}, d={list(full_dims)}) is out of reach of the n x n formulation "
                     f"({8.0 * n_full * n_full / 1e12:.0f} TB for Q); no extrapolation is given")}


def cpu_loss_baseline(batch=8192, d=512):
    """The reference's own CCALoss (oracle.losses.cca_loss_autograd: torch eigh + autograd, deep/objectives.py:61-102)
    forward + backward on the host, BASELINE configs[3] shape, median of three."""
    import numpy as np
    import torch

    from oracle import losses as ol

    torch.manual_seed(0)
    z1 = torch.randn(batch, d, requires_grad=True)
    z2 = (0.5 * z1.detach() + torch.randn(batch, d)).requires_grad_(True)
    times = []
    affinity, quota = host_cores()
    torch.set_num_threads(max(int(min(affinity, quota)) if quota else affinity, 1))
    for _ in range(3):
        z1.grad = z2.grad = None
        t0 = time.perf_counter()
        ol.cca_loss_autograd(z1, z2, 1e-6).backward()
        times.append(time.perf_counter() - t0)
    return {"metric": f"CPU CCALoss fwd+bwd (torch eigh autograd, batch {batch}, 2x{d}, fp32)", "ms": float(np.median(times)) * 1e3,
            "value": 1.0 / float(np.median(times)), "torch_threads": torch.get_num_threads()}


# ---------------------------------------------------------------------------------------------------------------
# extras (N = 1): the other BASELINE configurations and the second half of the metric
# ---------------------------------------------------------------------------------------------------------------
def loss_parity_gate(z1, z2, eps, loss, tol=1e-3, n_rows_check=2048):
    """Value and gradients of a timed CCALoss evaluation against the float64 closed form (oracle.losses).  Small batches:
    the closed form on the whole batch on the host.  Large ones (the metric shape): float64 moments of the SAME batch
    formed chunk by chunk on the device with torch (the comparator, not the product), the closed form evaluated on the
    host from those moments, the gradient compared on ``n_rows_check`` rows spread over the batch."""
    import numpy as np
    import torch

    from oracle import losses as ol

    n, d1, d2 = int(z1.shape[0]), int(z1.shape[1]), int(z2.shape[1])
    rep = {"tol": tol}
    if n * (d1 + d2) <= 8192 * 2048:
        l, g1, g2 = ol.cca_loss_closed_form(z1.detach().cpu().numpy(), z2.detach().cpu().numpy(), eps)
        rep["comparator"] = "oracle.losses.cca_loss_closed_form on the whole batch (host, float64)"
        e1 = float(np.linalg.norm(z1.grad.cpu().numpy() - g1) / np.linalg.norm(g1))
        e2 = float(np.linalg.norm(z2.grad.cpu().numpy() - g2) / np.linalg.norm(g2))
    else:
        D = d1 + d2
        G = torch.zeros(D, D, dtype=torch.float64, device=z1.device)
        sm = torch.zeros(D, dtype=torch.float64, device=z1.device)
        step = max(1, (1 << 28) // D)
        with torch.no_grad():
            for r0 in range(0, n, step):
                blk = torch.cat([z1[r0:r0 + step], z2[r0:r0 + step]], dim=1).double()
                G += blk.T @ blk
                sm += blk.sum(0)
                del blk
        l, Gamma, mean = ol.cca_loss_from_moments(G.cpu().numpy(), sm.cpu().numpy(), n, d1, d2, eps)
        del G
        rows = torch.arange(0, n, max(1, n // n_rows_check), device=z1.device)[:n_rows_check]
        with torch.no_grad():
            Z = torch.cat([z1[rows], z2[rows]], dim=1).double().cpu().numpy()
        gr = (Z - mean) @ Gamma
        rep["comparator"] = (f"oracle.losses.cca_loss_from_moments on float64 moments of the same batch; gradient on {len(rows)} rows")
        e1 = float(np.linalg.norm(z1.grad[rows].double().cpu().numpy() - gr[:, :d1]) / np.linalg.norm(gr[:, :d1]))
        e2 = float(np.linalg.norm(z2.grad[rows].double().cpu().numpy() - gr[:, d1:]) / np.linalg.norm(gr[:, d1:]))
    rep["loss"] = float(loss)
    rep["loss_ref"] = float(l)
    rep["loss_rel_err"] = abs(float(loss) - float(l)) / abs(float(l))
    rep["grad_rel_err"] = [e1, e2]
    rep["ok"] = bool(rep["loss_rel_err"] <= tol and e1 < tol and e2 < tol and np.isfinite(float(loss)))
    return rep


def dcca_extra(steps=20, warmup=3, batch=8192, d=512, label="BASELINE configs[3]", gate=True):
    """CCALoss forward + backward (ccz_pair_loss through the autograd Function), fp32 embeddings resident in HBM.
    Steps are enqueued back to back and the device is drained ONCE (the objective never synchronises the host): the
    reported time is wall-clock per step of that queue; ``ms_sync_each`` is the same loop with a host wait per step."""
    import numpy as np
    import torch

    from cca_zoo_amd.deep.objectives import CCALoss, check_async_errors

    torch.manual_seed(0)
    z1 = torch.randn(batch, d, device="cuda", requires_grad=True)
    z2 = torch.randn(batch, d, device="cuda")
    z2.add_(z1.detach(), alpha=0.5)
    z2.requires_grad_(True)
    obj = CCALoss(eps=1e-6)
    for _ in range(warmup):
        obj([z1, z2]).backward()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        z1.grad = None
        z2.grad = None
        t0 = time.perf_counter()
        obj([z1, z2]).backward()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt_sync = float(np.median(ts))
    # back-to-back: the way a training loop issues it
    reps = max(3, steps)
    t0 = time.perf_counter()
    for _ in range(reps):
        z1.grad = None
        z2.grad = None
        loss = obj([z1, z2])
        loss.backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    check_async_errors()
    # forward Gram n D (D + 1) with D = 2 d, backward (n x 2d) @ (2d x 2d)
    f_fwd = float(batch) * (2 * d) * (2 * d + 1)
    f_bwd = 2.0 * batch * (2 * d) * (2 * d)
    flops = f_fwd + f_bwd
    # arithmetic route of each half (ccz_loss_last_route): the split-bf16 route EXECUTES three bf16 products per algorithmic one,
    # so the roofline is stated on executed flops against the pipe each half ran on (a blended peak when the halves differ)
    from cca_zoo_amd import _backend

    r_fwd, r_bwd = _backend.handle_for([z1, z2]).loss_last_route()
    ex_fwd, pk_fwd = (3.0 * f_fwd, PEAK_TFLOPS["bf16"]) if r_fwd == "bf16x2" else (f_fwd, PEAK_TFLOPS["f32"])
    ex_bwd, pk_bwd = (3.0 * f_bwd, PEAK_TFLOPS["bf16"]) if r_bwd == "bf16x2" else (f_bwd, PEAK_TFLOPS["f32"])
    executed = ex_fwd + ex_bwd
    t_at_peak = ex_fwd / (pk_fwd * 1e12) + ex_bwd / (pk_bwd * 1e12)
    out = {"metric": f"DCCA CCALoss fwd+bwd/sec (batch {batch}, 2x{d}, fp32; {label})", "value": 1.0 / dt, "ms": dt * 1e3,
           "ms_sync_each": dt_sync * 1e3, "ms_min_sync_each": float(min(ts)) * 1e3, "tflops": flops / dt / 1e12, "flop": flops,
           "routes": {"forward_k1": r_fwd, "backward_product": r_bwd},
           "roofline": {"bound": "mfma", "achieved": executed / dt / 1e12, "peak": executed / t_at_peak / 1e12, "unit": "TFLOP/s",
                        "frac": t_at_peak / dt, "executed_flop": executed, "algorithmic_tflops": flops / dt / 1e12,
                        "note": "executed flops of the whole fwd+bwd (K1 + gradient GEMM; 3 bf16 products per algorithmic one on the "
                                "split route) over its wall time; peak = the dense MFMA peak of the pipe(s) those flops ran on"}}
    if gate:
        out["parity_gate"] = loss_parity_gate(z1, z2, 1e-6, loss.item())
    # stated context, never the target: what a user of the reference sees on THIS GPU -- its own loss (torch.linalg.eigh +
    # autograd: cca_zoo/deep/objectives.py:61-102 as restated by the oracle) on the same embeddings on cuda, warm, back to back
    try:
        from oracle import losses as ol

        a = z1.detach().clone().requires_grad_(True)
        b = z2.detach().clone().requires_grad_(True)
        for _ in range(2):
            ol.cca_loss_autograd(a, b, 1e-6).backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            a.grad = None
            b.grad = None
            ol.cca_loss_autograd(a, b, 1e-6).backward()
        torch.cuda.synchronize()
        out["torch_gpu_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        out["torch_gpu_note"] = "the oracle's restatement of the reference loss (torch eigh + autograd) on cuda: context, not the target"
    except Exception as e:                                   # a comparator must never take the measurement down
        out["torch_gpu_ms"] = None
        out["torch_gpu_note"] = f"comparator failed: {type(e).__name__}"
    return out


def training_step_extra(batch=8192, d_in=784, hidden=1024, d_out=512, steps=30, warmup=5):
    """One DCCA ``training_step`` (the reference: cca_zoo/deep/_base.py:78-104 -- encoders forward, the objective,
    backward) with MLP encoders as in the reference's user guide (docs/user-guide/deep.md:225-232), timed (a) with
    CCALoss, (b) with a trivial stand-in objective (the encoders alone), and (c) the objective alone on the same
    embeddings.  If the objective drained the queue, (a) would exceed (b) + (c); with device-side stream hand-over it
    does not."""
    import torch
    import torch.nn as nn

    from cca_zoo_amd.deep.objectives import CCALoss

    torch.manual_seed(0)

    def mlp():
        return nn.Sequential(nn.Linear(d_in, hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU(), nn.Linear(hidden, d_out)).cuda()

    e1, e2 = mlp(), mlp()
    x1 = torch.randn(batch, d_in, device="cuda")
    x2 = torch.randn(batch, d_in, device="cuda") + 0.5 * x1
    obj = CCALoss(eps=1e-6)
    params = list(e1.parameters()) + list(e2.parameters())

    def step(with_loss):
        for p in params:
            p.grad = None
        z1, z2 = e1(x1), e2(x2)
        loss = obj([z1, z2]) if with_loss else (z1.square().mean() - (z1[:, :d_out] * z2).mean())
        loss.backward()

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    with_ms = timed(lambda: step(True))
    without_ms = timed(lambda: step(False))
    z1 = e1(x1).detach().requires_grad_(True)
    z2 = e2(x2).detach().requires_grad_(True)

    def loss_only():
        z1.grad = None
        z2.grad = None
        obj([z1, z2]).backward()

    alone_ms = timed(loss_only)
    return {"metric": f"DCCA training_step (two MLP encoders {d_in}-{hidden}-{hidden}-{d_out}, batch {batch}, fp32): fwd + CCALoss + bwd",
            "step_ms_with_loss": with_ms, "step_ms_encoders_only": without_ms, "loss_alone_ms": alone_ms,
            "loss_increment_ms": with_ms - without_ms,
            "note": "loss_increment ~ loss_alone means the objective adds its own kernel time and no queue drain"}


def transform_extra(h, views, model):
    """``transform`` of the first timed view (n x d fp32 -> n x k): HBM-bound by its algorithmic bytes n d 4.
    The fp32 kernel (what ``auto`` runs) and the opt-in split projection (``ccz_k1_route(bf16x2)``), each against a float64
    product on 65536 rows."""
    import ctypes as C

    import numpy as np
    import torch

    from cca_zoo_amd import _backend

    X = views[0]
    n, d = int(X.shape[0]), int(X.shape[1])
    W = torch.as_tensor(np.ascontiguousarray(model.weights_[0], dtype=np.float64), device=X.device)
    mean = torch.as_tensor(np.asarray(model.means_[0], dtype=np.float64), device=X.device)
    k = int(W.shape[1])
    out = torch.empty((n, k), dtype=torch.float32, device=X.device)
    ref = (X[:65536].double() - mean) @ W
    res = {"metric": f"transform of one {n} x {d} fp32 view onto k = {k} directions", "bytes": float(n) * d * 4}
    prev = h.k1_route(None)
    try:
        for route in ("auto", "bf16x2"):
            h.k1_route(route)
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                h.check(h.lib.ccz_transform(h.raw, _backend.F32, C.c_void_p(X.data_ptr()), n, d, X.stride(0), C.c_void_p(mean.data_ptr()),
                                            C.c_void_p(W.data_ptr()), k, C.c_void_p(out.data_ptr()), out.stride(0)))
                h.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            ms = float(np.median(ts[1:]))
            err = float((out[:65536].double() - ref).norm() / ref.norm())
            res["fp32_kernel" if route == "auto" else "split_projection_opt_in"] = {
                "ms": ms, "rel_err_vs_float64": err,
                "roofline": {"bound": "hbm", "achieved": n * d * 4 / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                             "frac": n * d * 4 / (ms * 1e-3) / 1e9 / 8000.0}}
        # k <= 32 directions (the usual latent dimension of the linear models): one 32-wide MFMA column tile instead of two --
        # half the fp32 matrix-pipe work, which is what holds the k = 64 form below the HBM rate
        h.k1_route("auto")
        k2 = min(32, k)
        W2 = W[:, :k2].contiguous()
        out2 = torch.empty((n, k2), dtype=torch.float32, device=X.device)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h.check(h.lib.ccz_transform(h.raw, _backend.F32, C.c_void_p(X.data_ptr()), n, d, X.stride(0), C.c_void_p(mean.data_ptr()),
                                        C.c_void_p(W2.data_ptr()), k2, C.c_void_p(out2.data_ptr()), out2.stride(0)))
            h.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        ms = float(np.median(ts[1:]))
        res[f"fp32_kernel_k{k2}"] = {
            "ms": ms, "rel_err_vs_float64": float((out2[:65536].double() - ref[:, :k2]).norm() / ref[:, :k2].norm()),
            "roofline": {"bound": "hbm", "achieved": n * d * 4 / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": n * d * 4 / (ms * 1e-3) / 1e9 / 8000.0}}
    finally:
        h.k1_route(prev)
    return res


def timed_fit(make_model, views, runs):
    import numpy as np
    import torch

    import gc

    ts, solves, last = [], [], None
    # interpreter housekeeping out of the timed region, as in the headline loop (and as timeit does): after the gates'
    # host work a generation-2 collection of the sklearn + torch + scipy heap costs ~30 ms and lands inside a "solve"
    gc.collect()
    gc.disable()
    # the gate of the PREVIOUS configuration ends with seconds of multi-threaded host BLAS: under a cgroup CPU quota that
    # leaves a throttling debt, and a launch chain of ~400 dispatches issued by a throttled host thread is what made one fit
    # in three of an extra 10-20 ms slower (solve_ms_runs [32.2, 25.6, 32.2] in BENCH_r04).  Three quota periods of rest.
    time.sleep(0.3)
    try:
        for _ in range(runs + 1):            # first run warms allocator pools / code objects and is dropped
            t0 = time.perf_counter()
            last = make_model().fit(views)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            solves.append(last.timings_["solve_ms"])
    finally:
        gc.enable()
    last.timings_["solve_ms_median"] = float(np.median(solves[1:]))
    last.timings_["solve_ms_all"] = [round(float(x), 2) for x in solves[1:]]
    return float(np.median(ts[1:])), last


def config_extras(info, gates=True):
    """BASELINE configs[1], [2] and [4] (the last at the largest n one GPU holds), the metric shape in float64 and the
    metric shape on data far from zero (mean = 10 sigma: the pilot-shifted K1): fit time, K1 rate, solve time -- each
    behind its own parity gate (``solution_gate``); a configuration whose gate fails is reported as failed, without
    numbers.  Each configuration draws its own views and frees them."""
    import numpy as np
    import torch

    from cca_zoo_amd import _backend
    from cca_zoo_amd.datasets import JointData
    from cca_zoo_amd.linear import CCA, GCCA, MCCA, rCCA

    h = _backend.default_handle()
    out = {}

    def run(tag, label, dims, n, k, tdt, make_model, est, c, runs=2, offset=0.0):
        free, _ = torch.cuda.mem_get_info()
        need = n * sum(dims) * (4 if tdt == torch.float32 else 8)
        if need * 1.04 + 14e9 > free:
            out[tag] = {"config": label, "skipped": f"needs {need / 1e9:.0f} GB of HBM, {free / 1e9:.0f} GB free"}
            return
        jd = JointData(n_views=len(dims), n_samples=1, latent_dimensions=k, n_features=list(dims), random_state=1,
                       latent_scales=list(np.linspace(2.0, 0.5, k)))
        views = jd.sample_device(device="cuda", dtype=tdt, n_samples=n, seed=DATA_SEED + 1)
        if offset:
            for v in views:                       # every feature sits `offset` standard deviations away from zero
                v.add_(offset * float(v[:4096].std()))
        ms, model = timed_fit(make_model, views, runs)
        g_ms = h.moments_last_ms()[0]
        D = sum(dims)
        flop = float(n) * D * (D + 1)
        kind = "f32" if tdt == torch.float32 else "f64"
        res = {"config": label, "fit_ms": ms, "fits_per_s": 1e3 / ms, "gram_ms": g_ms, "solve_ms": model.timings_["solve_ms_median"], "solve_ms_runs": model.timings_["solve_ms_all"],
               **k1_rates(h, flop, g_ms, kind),
               "dtype": kind, "n": n, "pilot_shifted_k1": bool(h.moments_last_pilot()),
               "score_top": float(np.asarray(model.score(views))[0])}
        if gates:
            try:
                gate = solution_gate(model, views, est, c, jd=None if offset else jd, seed=DATA_SEED + 1, inertia=D <= 8192)
            except Exception as e:                                            # a gate that cannot run does not pass
                gate = {"ok": False, "error": f"{type(e).__name__}: {e}"}
            if gate["ok"]:
                res["parity_gate"] = gate
            else:
                res = {"config": label, "failed_parity_gate": gate}
        out[tag] = res
        del views, model
        torch.cuda.empty_cache()
        h.pool_trim()                         # the split route's planes and partial tiles (tens of GB) go back before the next draw

    run("c2_rcca", "configs[1]: rCCA n=100k, 2x1024, k=32, float32", (1024, 1024), 100_000, 32, torch.float32,
        lambda: rCCA(latent_dimensions=32, c=0.1), "rcca", 0.1, runs=5)
    run("c3_mcca", "configs[2]: MCCA 4x2048, n=1e6, k=64, float32 (one GPU holds all rows)", (2048,) * 4, 1_000_000, 64,
        torch.float32, lambda: MCCA(latent_dimensions=64), "mcca", 0.0, runs=3)
    run("ns_offset", "metric shape on off-centre data (every column mean = 10 sigma): CCA n=1e6, 2x4096, k=64, float32",
        (4096, 4096), 1_000_000, 64, torch.float32, lambda: CCA(latent_dimensions=64), "rcca", 0.0, runs=3, offset=10.0)
    run("ns_f64", "metric shape in float64: CCA n=1e6, 2x4096, k=64", (4096, 4096), 1_000_000, 64, torch.float64,
        lambda: CCA(latent_dimensions=64), "rcca", 0.0, runs=3)
    # configs[4]: the largest n ONE GPU holds, chosen from what is free now (views n x 16384 float64 + 2.1 GB of moments +
    # ~30 GB for the solver's D x D scratch blocks, the generator's chunks and allocator slack), in steps of 65536 rows, at most the configuration's own 2e6
    free, _ = torch.cuda.mem_get_info()
    n5 = int(min(2_000_000, max(65536, ((free - 32e9) / 1.03 / (16384 * 8)) // 65536 * 65536)))
    for n_try in (n5, 1_000_000):
        try:
            run("c5_gcca", f"configs[4] at the largest n one GPU holds: GCCA d=[4096,4096,8192], n={n_try} (of 2e6; chosen from "
                f"{free / 1e9:.0f} GB free), k=128, float64", (4096, 4096, 8192), n_try, 128, torch.float64,
                lambda: GCCA(latent_dimensions=128), "gcca", 0.0)
            break
        except (torch.cuda.OutOfMemoryError, MemoryError) as e:                # (libccz's pool reports CCZ_ENOMEM as MemoryError)
            out["c5_gcca"] = {"config": f"configs[4], n={n_try}", "skipped": f"out of HBM: {e}"[:300]}
            torch.cuda.empty_cache()
    return out


def sharded_dcca_extra(n_local, d, world, steps=2, warmup=1):
    """DCCA CCALoss fwd+bwd on a batch of n rows sharded over the ranks (metric shape): K1 per shard, one packed
    all-reduce, replicated d x d solve, local gradient GEMM.  Every rank calls this; returns seconds per step
    (max over ranks)."""
    import torch
    import torch.distributed as dist

    from cca_zoo_amd import row_sharded
    from cca_zoo_amd.deep.objectives import CCALoss

    torch.manual_seed(100 + dist.get_rank())
    z1 = torch.randn(n_local, d, device="cuda", requires_grad=True)
    z2 = torch.randn(n_local, d, device="cuda")
    z2.add_(z1.detach(), alpha=0.5)
    z2.requires_grad_(True)
    obj = CCALoss(eps=1e-6)
    for _ in range(warmup):
        with row_sharded():
            obj([z1, z2]).backward()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        z1.grad = None
        z2.grad = None
        with row_sharded():
            obj([z1, z2]).backward()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([(time.perf_counter() - t0) / steps], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def weights_agreement(W, W_ref, vals, tol, sep_factor=100.0):
    """Per-column relative error after sign alignment where the neighbouring canonical correlations are separated by more
    than sep_factor x tol (default 100); otherwise the residual of W_ref in span(W) -- eigenvectors inside a cluster of
    nearly equal correlations are only defined up to a rotation of the cluster, in ANY solver at this precision."""
    import numpy as np

    vals = np.asarray(vals, dtype=np.float64)
    gap = np.full(len(vals), np.inf)
    if len(vals) > 1:
        dv = np.abs(np.diff(vals))
        gap[:-1] = np.minimum(gap[:-1], dv)
        gap[1:] = np.minimum(gap[1:], dv)
    sep = gap > sep_factor * tol * max(abs(vals[0]), 1e-300)
    col, sub = 0.0, 0.0
    for w, r in zip(W, W_ref):
        w, r = np.asarray(w, dtype=np.float64), np.asarray(r, dtype=np.float64)
        sgn = np.sign(np.sum(w * r, axis=0))
        e = np.linalg.norm(w * sgn - r, axis=0) / np.linalg.norm(r, axis=0)
        if sep.any():
            col = max(col, float(e[sep].max()))
        coef, *_ = np.linalg.lstsq(w, r, rcond=None)
        sub = max(sub, float(np.linalg.norm(r - w @ coef) / np.linalg.norm(r)))
    return {"separated_columns": int(sep.sum()), "of": int(len(vals)), "max_col_rel_err_separated": col,
            "subspace_residual": sub, "min_rel_gap": float(gap.min() / max(abs(vals[0]), 1e-300)), "ok": bool(col < tol and sub < tol)}


def evd_extra(sizes=(512, 1024, 2048, 4096)):
    """The dense symmetric EVD behind ``svd_whiten`` / ``_inv_sqrtm`` / ``_BatchWhiten`` (``ccz_syevj``, blocked Jacobi of
    csrc/evd_block.hip; reference: numpy.linalg.svd / torch.linalg.eigh at cca_zoo/_utils/_linalg.py:28, deep/objectives.py:19)
    and the thin SVD behind ``np.linalg.svd(cross_cov)`` (``ccz_gesvj``; cca_zoo/linear/_rcca.py:97) at the seams' sizes:
    warm milliseconds, sweeps, the nominal 9 d^3 rate, and the gate -- ||A V - V L|| / ||A|| and ||V'V - I|| (Frobenius,
    formed with torch float64 as the comparator) below 1e-11."""
    import ctypes as C

    import numpy as np
    import torch

    from cca_zoo_amd import _backend

    h = _backend.default_handle()
    out, ok = {}, True
    g = torch.Generator(device="cuda").manual_seed(5)
    for d in sizes:
        X = torch.randn(3 * d, d, dtype=torch.float64, device="cuda", generator=g) * torch.linspace(2.0, 0.05, d, dtype=torch.float64, device="cuda")
        A = (X.T @ X / (3 * d - 1)).contiguous()
        w = torch.empty(d, dtype=torch.float64, device="cuda")
        V = torch.empty(d, d, dtype=torch.float64, device="cuda")
        sw = C.c_int(0)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h.check(h.lib.ccz_syevj(h.raw, C.c_void_p(A.data_ptr()), d, C.c_void_p(w.data_ptr()), C.c_void_p(V.data_ptr()), C.byref(sw)))
            h.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        ms = float(min(ts[1:]))
        resid = float(torch.linalg.norm(A @ V.T - V.T * w) / torch.linalg.norm(A))
        orth = float(torch.linalg.norm(V @ V.T - torch.eye(d, dtype=torch.float64, device="cuda")))
        good = resid < 1e-11 and orth < 1e-11 and bool(torch.all(w[:-1] >= w[1:]))
        ok = ok and good
        # stated context, never the target: what the reference itself runs on this GPU at this seam -- torch.linalg.eigh
        # (hipSOLVER through PyTorch-ROCm; cca_zoo/deep/objectives.py:19) on the same matrix, warm
        eigh_ms = None
        try:
            torch.linalg.eigh(A)
            torch.cuda.synchronize()
            te = []
            for _ in range(2):
                t0 = time.perf_counter()
                torch.linalg.eigh(A)
                torch.cuda.synchronize()
                te.append((time.perf_counter() - t0) * 1e3)
            eigh_ms = float(min(te))
        except Exception:
            eigh_ms = None
        out[f"syev_{d}"] = {"ms": ms, "first_call_ms": ts[0], "sweeps": sw.value, "nominal_tflops_9d3": 9.0 * d ** 3 / (ms * 1e-3) / 1e12,
                            "residual": resid, "orthogonality": orth, "torch_eigh_ms": eigh_ms}
        del X, A, V
    for p, q in ((1024, 1024), (4096, 1024)):
        A = torch.randn(p, q, dtype=torch.float64, device="cuda", generator=g)
        r = min(p, q)
        U = torch.empty(p, r, dtype=torch.float64, device="cuda")
        sg = torch.empty(r, dtype=torch.float64, device="cuda")
        Vt = torch.empty(r, q, dtype=torch.float64, device="cuda")
        sw = C.c_int(0)
        ts = []
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h.check(h.lib.ccz_gesvj(h.raw, C.c_void_p(A.data_ptr()), p, q, C.c_void_p(U.data_ptr()), C.c_void_p(sg.data_ptr()),
                                    C.c_void_p(Vt.data_ptr()), C.byref(sw)))
            h.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        rec = float(torch.linalg.norm(U * sg @ Vt - A) / torch.linalg.norm(A))
        orth = float(max(torch.linalg.norm(U.T @ U - torch.eye(r, dtype=torch.float64, device="cuda")),
                         torch.linalg.norm(Vt @ Vt.T - torch.eye(r, dtype=torch.float64, device="cuda"))))
        good = rec < 1e-11 and orth < 1e-10
        ok = ok and good
        svd_ms = None
        try:
            torch.linalg.svd(A, full_matrices=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            torch.linalg.svd(A, full_matrices=False)
            torch.cuda.synchronize()
            svd_ms = (time.perf_counter() - t0) * 1e3
        except Exception:
            svd_ms = None
        out[f"gesv_{p}x{q}"] = {"ms": float(min(ts)), "sweeps": sw.value, "reconstruction": rec, "orthogonality": orth, "torch_svd_ms": svd_ms}
        del A, U, Vt
    torch.cuda.empty_cache()
    out["parity_gate"] = {"ok": bool(ok)}
    return out


def host_inputs_extra(views, k, rows=262_144):
    """The scikit-learn surface's NATIVE input: ``CCA(k).fit`` on HOST NumPy views -- pageable (what
    ``np.asarray`` gives: libccz packs row chunks into pinned bounce buffers on host threads || DMA || K1) and pinned
    (``torch.Tensor.pin_memory().numpy()``: the DMA reads the caller's pages directly) -- at the metric shape's widths on
    the first ``rows`` rows of the timed views.  Reported: inclusive seconds per fit, the PCIe rate that implies, and the
    rate of a plain pinned copy of the same bytes on this box (the link's achievable peak).  Gate: the weights equal those
    of the HBM-resident fit of the same rows (float32 bar 1e-3 per column) and the scores agree."""
    import numpy as np
    import torch

    from cca_zoo_amd.linear import CCA

    rows = min(rows, int(views[0].shape[0]))
    dev_views = [v[:rows] for v in views]
    ref = CCA(latent_dimensions=k).fit(dev_views)
    ref_score = np.asarray(ref.score(dev_views), dtype=np.float64)
    pageable = [v.cpu().numpy() for v in dev_views]
    nbytes = float(sum(p.nbytes for p in pageable))
    pinned_t = [torch.from_numpy(p).pin_memory() for p in pageable]
    pinned = [t.numpy() for t in pinned_t]
    # the link: a plain pinned -> HBM copy of the same bytes
    dst = [torch.empty_like(v) for v in dev_views]
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for d, t in zip(dst, pinned_t):
            d.copy_(t, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    del dst
    link = nbytes / best / 1e9
    out = {"rows": rows, "bytes": nbytes, "pinned_copy_GBps": link}
    ok = True
    for name, hv in (("pageable", pageable), ("pinned", pinned)):
        m = CCA(latent_dimensions=k).fit(hv)             # warm-up: bounce buffers, staging blocks
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            m = CCA(latent_dimensions=k).fit(hv)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        agree = weights_agreement(m.weights_, ref.weights_, ref.singular_values_, 1e-3)
        sc = float(np.abs(np.asarray(m.score(hv), dtype=np.float64) - ref_score).max())
        good = agree["ok"] and sc < 1e-4
        ok = ok and good
        out[name] = {"fit_s": t, "fits_s": [round(x, 4) for x in ts], "inclusive_GBps": nbytes / t / 1e9,
                     "frac_of_pinned_copy": nbytes / t / 1e9 / link, "weights_vs_hbm_resident_fit": agree, "score_diff": sc,
                     "fit_per_s_at_n_1e6_extrapolated": 1.0 / (t * 1e6 / rows)}
    out["parity_gate"] = {"ok": bool(ok)}
    return out


def grid_extra(views, k, fit_ms):
    """GridSearchCV over 8 ridge values x 5 folds (+ refit) on the SAME views from one pass
    over the data (moments per fold, training moments by subtraction); the reference refits 41 times."""
    import torch

    from cca_zoo_amd.linear import rCCA
    from cca_zoo_amd.model_selection import GridSearchCV

    grid = {"c": [1e-4, 1e-3, 1e-2, 0.05, 0.1, 0.3, 0.6, 0.9]}
    t0 = time.perf_counter()
    gs = GridSearchCV(rCCA(latent_dimensions=k), grid, cv=5).fit(views)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"metric": "GridSearchCV(rCCA, 8 values of c, cv=5, refit) on the bench views", "seconds": dt,
            "moments_pass_s": gs.moments_pass_time_, "mean_solve_ms": float(gs.cv_results_["mean_fit_time"].mean() * 1e3),
            "refit_per_setting_equivalent_s": 41 * fit_ms * 1e-3, "best_params": gs.best_params_,
            "best_score": gs.best_score_}


def gram_traffic(dtype, D, n_local):
    """HBM / fabric bytes of one K1 launch from the newest committed PMC pass for this dtype and width (FETCH_SIZE x 2
    gfx950 correction + WRITE_SIZE per row, measured at a smaller n on the same kernel and shape; linear in the rows)."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gram_traffic*.json"))):
        try:
            with open(path) as f:
                tj = json.load(f)
        except Exception:
            continue
        if tj.get("D") == D and tj.get("dtype", "f32") == dtype:
            best = (tj, os.path.basename(path))
    if best is None:
        return None, None
    return best[0]["bytes_per_row"] * n_local, f"profiles/{best[1]} (PMC pass at n={best[0].get('n')}, scaled by rows)"


def relaunch(n_gpus):
    """``python bench.py --gpus N`` without a launcher: start the N ranks ourselves (one process per GPU under
    ``torch.distributed.run``, rendezvous on 127.0.0.1, a free port) and pass their output through -- rank 0 prints the ONE
    JSON line.  The driver's own ``python -m torch.distributed.run ... bench.py --gpus N`` form sets WORLD_SIZE and never
    comes here."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, n_gpus))))
    return subprocess.call(cmd, env=env)


def launch_test():
    """The launcher logic without a GPU: every rank joins a gloo group, one all-reduce, rank 0 prints ONE JSON line."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29519")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    ranks = dist.get_world_size()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_test": True, "n_gpus": world, "ranks": ranks, "sum": float(t.item())}), flush=True)


def main():
    # idle BLAS / OpenMP workers should sleep, not spin, between the host-side gates and the next timed launch chain
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    os.environ.setdefault("OPENBLAS_THREAD_TIMEOUT", "10")
    os.environ.setdefault("GOMP_SPINCOUNT", "0")
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(a.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')} "
                 "(launch N ranks for --gpus N, or let bench.py start them)")
    if a.launch_test:
        return launch_test()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    one_gpu = a.transport == "gloo-staged"                   # test transport: every rank on cuda:0, collectives staged over gloo
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    os.environ["CCZ_DEVICE"] = str(local)
    # One BLAS / OpenMP thread per usable core for the WHOLE process (the gates run LAPACK on the host): the default is one
    # per logical CPU of the box (256) under a 16-cpu cgroup quota -- spinning workers burn the quota, the cgroup is
    # throttled, and the launch chain of the NEXT timed solve stalls for tens of milliseconds (first bench of round 4:
    # the extras' solves at 28-50 ms instead of 14 after the headline gate's 8192 x 8192 LDL' factorization).
    affinity, quota = host_cores()
    # (two cores stay free for the launching thread and libccz's helpers: a full set of spinning BLAS workers next to
    # the main thread puts the cgroup over its quota again)
    n_thr = max(1, (int(min(affinity, quota) if quota else affinity) - 2) // max(1, world))
    torch.set_num_threads(n_thr)
    try:
        from threadpoolctl import threadpool_limits

        _blas_cap = threadpool_limits(limits=n_thr)      # kept for the life of the process
    except Exception:
        _blas_cap = None
    # CCZ_BENCH_FORCE_SHARDED=1 runs the N > 1 code path (process group, row_sharded fits, sharded loss) with one rank
    distributed = world > 1 or bool(os.environ.get("CCZ_BENCH_FORCE_SHARDED"))
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if one_gpu:
            # gloo prints its "[Gloo] Rank r is connected ..." lines on the C stdout: the contract is ONE JSON line there
            import ctypes

            sys.stdout.flush()
            saved_fd = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("gloo", rank=rank, world_size=world)
                dist.barrier()
                ctypes.CDLL(None).fflush(None)
            finally:
                os.dup2(saved_fd, 1)
                os.close(saved_fd)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from cca_zoo_amd import _backend, row_sharded, shard_bounds
    from cca_zoo_amd.datasets import JointData
    from cca_zoo_amd.linear import CCA

    h = _backend.default_handle(local)
    h.k1_route(a.k1_route)
    info = h.device_info()
    lo, hi = shard_bounds(a.n, rank, world)
    n_local = hi - lo
    tdt = torch.float32 if a.dtype == "f32" else torch.float64
    jd = JointData(n_views=2, n_samples=a.n, latent_dimensions=a.k, n_features=[a.d, a.d],
                   signal_to_noise=1.0, random_state=0, latent_scales=list(np.linspace(2.0, 0.5, a.k)))
    # ONE global data set: this rank draws rows [lo, hi) of it, whatever the number of ranks
    views = jd.sample_device(device=f"cuda:{local}", dtype=tdt, n_samples=n_local, seed=DATA_SEED, row0=lo)
    torch.cuda.synchronize()
    model = CCA(latent_dimensions=a.k)
    comm = None
    if distributed and a.transport == "ccz":
        # libccz's own communicator: rank 0's id travels over the process group that the gates use anyway
        from cca_zoo_amd import _dist

        idt = torch.zeros(128, dtype=torch.uint8, device=f"cuda:{local}")
        if rank == 0:
            idt = torch.tensor(list(h.comm_unique_id()), dtype=torch.uint8, device=f"cuda:{local}")
        dist.broadcast(idt, 0)
        comm = _dist.CczComm(h, bytes(idt.cpu().tolist()), world, rank)

    def step():
        if distributed:
            with row_sharded(group=comm):
                model.fit(views)
        else:
            model.fit(views)

    gram_ms, colsum_ms, solve_ms, allreduce_ms, exchange_events = [], [], [], [], []
    stage_ms = []                                         # split-bf16 route: (split, mfma, reduce) of every timed step
    from cca_zoo_amd import _moments

    _moments.TIME_EXCHANGE = distributed                  # event pairs around the two parts of the exchange (no host waits)
    # Process warm-up that is not a property of the step: allocator pools, code-object loads and whatever else makes
    # the first two or three fits of a process 15-30 ms slower.  Two untimed fits during set-up, in
    # addition to the W warm-up steps the caller asks for; reported as config.setup_fits.
    SETUP_FITS = 2
    for _ in range(SETUP_FITS):
        step()
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    # interpreter housekeeping out of the timed region, as timeit does: a generation-2 collection of the
    # (large, sklearn + torch) heap costs ~30 ms and used to land in a timed step
    import gc

    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(a.steps):
        ts = time.perf_counter()
        step()
        g, cs = h.moments_last_ms()
        gram_ms.append(g)
        colsum_ms.append(cs)
        stage_ms.append(h.moments_last_route())
        solve_ms.append(model.timings_["solve_ms"])
        allreduce_ms.append(model.timings_["allreduce_ms"])
        step_ms.append((time.perf_counter() - ts) * 1e3)
        if distributed:
            exchange_events.append(_moments.exchange_events())
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
        dist_all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / a.steps * 1e3
    # the exchange on the device time line: head = pack -> first all-reduce done; tail = first done -> second unpacked
    # (the solve's factorization overlaps the tail; what it could not hide is inside phases_ms.solve)
    exchange_split = None
    evs = [e for e in exchange_events if e]
    if evs:
        exchange_split = {"head_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in evs])),
                          "tail_ms": float(np.mean([e[1].elapsed_time(e[2]) for e in evs if e[2] is not None] or [0.0])),
                          "host_wait_for_head_ms": float(np.mean(allreduce_ms))}
    k1_ms_per_rank = [float(np.mean(gram_ms))]
    rccl_ranks = 1
    ccz_comm_ranks = None
    if comm is not None:
        ccz_comm_ranks = int(h.comm_info()[0])
    if distributed:
        rccl_ranks = dist.get_world_size()
        gt = torch.zeros(world, dtype=torch.float64, device=f"cuda:{local}")
        gt[rank] = float(np.mean(gram_ms))
        dist_all_reduce(gt)
        k1_ms_per_rank = [float(x) for x in gt.cpu().tolist()]

    # ---- parity gate on the views that were timed (every rank takes part: transform / score all-reduce) ----
    gate = check_fit_properties(model, views, jd, seed=DATA_SEED, row0=lo, sharded=distributed)
    if not gate["ok"]:
        sys.stderr.write("bench.py: PARITY GATE FAILED on the timed model: " + json.dumps(gate) + "\n")
        if distributed:
            dist.destroy_process_group()
        sys.exit(3)

    sharded_loss_s = None
    if distributed and (a.with_sharded_dcca or os.environ.get("CCZ_BENCH_FORCE_SHARDED")) and 8.0 * n_local * a.d * 4 < 150e9:
        del views
        torch.cuda.empty_cache()
        views = None
        sharded_loss_s = sharded_dcca_extra(n_local, a.d, world)

    if rank == 0:
        D = 2 * a.d
        flop = float(n_local) * D * (D + 1)                    # algorithmic flops of ONE Gram launch (this rank)
        g_ms = float(np.mean(gram_ms))
        route = stage_ms[-1][0] if stage_ms else ("fp32" if a.dtype == "f32" else "fp64")
        if route == "bf16x2":
            # split-bf16 route: the dominant kernel is k_gram_bf16x2, which EXECUTES 3 bf16 products per algorithmic one
            # (hi'hi + hi'mid + mid'hi in one accumulator).  achieved = 3 x F_gram / that kernel's own HIP-event time against
            # the dense bf16 peak; the algorithmic rate of the WHOLE K1 (split pass + MFMA kernel + reduce) stands beside it.
            kernel, kernel_ms = "k_gram_bf16x2", float(np.mean([sm[2] for sm in stage_ms]))
            achieved = 3.0 * flop / (kernel_ms * 1e-3) / 1e12
            peak = PEAK_TFLOPS["bf16"]
            traffic, traffic_src = gram_traffic("bf16x2", D, n_local)
            roof_extra = {"executed_flop_per_launch": 3.0 * flop, "algorithmic_tflops": flop / (g_ms * 1e-3) / 1e12,
                          "algorithmic_tflops_kernel_only": flop / (kernel_ms * 1e-3) / 1e12, "k1_ms": g_ms,
                          "k1_stages_ms": {"split": float(np.mean([sm[1] for sm in stage_ms])), "mfma": kernel_ms,
                                           "reduce": float(np.mean([sm[3] for sm in stage_ms]))},
                          "arithmetic": "fp32 rows as two bf16 planes (x - pilot = hi + mid); 3 bf16 MFMAs per fp32 product, fp32 "
                                        "accumulation per <= 16384-row chunk, fp64 across chunks; diag(sum mid^2) added back exactly"}
        else:
            kernel, kernel_ms = ("k_gram_f32_fifo" if a.dtype == "f32" else "k_gram_f64_fifo"), g_ms
            achieved = flop / (g_ms * 1e-3) / 1e12
            peak = PEAK_TFLOPS[a.dtype]
            traffic, traffic_src = gram_traffic(a.dtype, D, n_local)
            roof_extra = {}
        out = {
            "metric": "CCA fit()/sec at n=1e6 d=4096 k=64",
            "value": 1e3 / ms_per_step, "unit": "fit/s",
            "n_gpus": world, "rccl_ranks": (0 if one_gpu else rccl_ranks), "ranks_on_one_gpu": (world if one_gpu else None),
            "transport": (a.transport if distributed else None),
            "ccz_comm_ranks": ccz_comm_ranks, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "step_ms": [round(x, 2) for x in step_ms],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic (JointData latent-variable model, counter-based generator, drawn in HBM)",
            "config": {"workload": f"CCA(latent_dimensions={a.k}).fit on JointData n={a.n}, 2 views x {a.d}, {a.dtype}; "
                                   f"rows sharded over {world} GPU(s)", "n": a.n, "d": a.d, "k": a.k, "setup_fits": SETUP_FITS,
                       "device": info["name"], "arch": info["arch"], "compute_units": info["compute_units"]},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel, "k1_route": route,
                         "kernel_ms": kernel_ms, "flop_per_launch": flop,
                         "bytes_per_launch": float(n_local) * D * (4 if a.dtype == "f32" else 8),
                         "gram_share_of_step": g_ms / ms_per_step, **roof_extra},
            "phases_ms": {"gram": g_ms, "gram_per_rank": k1_ms_per_rank, "colsum": float(np.mean(colsum_ms)),
                          "allreduce": float(np.mean(allreduce_ms)), "allreduce_split": exchange_split,
                          "solve": float(np.mean(solve_ms)), "solve_min": float(np.min(solve_ms))},
            "parity_gate": gate,
        }
        extra = {}
        if sharded_loss_s is not None:
            extra["dcca_loss_metric_shape_sharded"] = {
                "metric": f"DCCA CCALoss fwd+bwd/sec (batch {a.n} sharded over {world} GPUs, 2x{a.d}, fp32)",
                "value": 1.0 / sharded_loss_s, "ms": sharded_loss_s * 1e3}
        if world == 1 and not a.no_extras and views is not None:
            gates = not a.no_gates

            def gated(name, res):
                """An extra whose parity gate failed is dropped (reported as failed, without numbers)."""
                g = res.get("parity_gate")
                if gates and g is not None and not g["ok"]:
                    sys.stderr.write(f"bench.py: parity gate of extra '{name}' FAILED: {json.dumps(res)[:4000]}\n")
                    extra[name] = {"metric": res.get("metric"), "failed_parity_gate": g}
                    return False
                extra[name] = res
                return True

            only = set(x for x in a.only.split(",") if x)
            want = lambda tag: not only or tag in only
            if a.dtype == "f32" and want("routes"):
                extra["k1_routes"] = k1_routes_extra(h, views, lambda: CCA(latent_dimensions=a.k))
            if not a.no_dcca and want("dcca"):
                gated("dcca_loss", dcca_extra(gate=gates))
                extra["dcca_training_step"] = training_step_extra()
            if a.dtype == "f32" and want("transform"):
                extra["transform"] = transform_extra(h, views, model)
            if want("grid"):
                extra["grid_search"] = grid_extra(views, a.k, ms_per_step)
            if want("host"):
                try:
                    gated("host_inputs", host_inputs_extra(views, a.k))
                except (MemoryError, RuntimeError) as e:          # a host without 17 GB to spare for the two copies
                    extra["host_inputs"] = {"skipped": f"{type(e).__name__}: {e}"[:300]}
            del views
            views = None
            torch.cuda.empty_cache()
            if not a.no_dcca and want("metric_loss") and a.n * a.d * 4 * 4 < 200e9:   # z1, z2 and their gradients must fit in HBM
                gated("dcca_loss_metric_shape", dcca_extra(steps=3, warmup=1, batch=a.n, d=a.d, label="metric shape", gate=gates))
                torch.cuda.empty_cache()
            if want("configs"):
                extra["configs"] = config_extras(info, gates=gates)
            if want("evd"):
                gated("dense_evd", evd_extra())
        if extra:
            out["extra"] = extra
        if "value" in extra.get("dcca_loss_metric_shape", {}):
            # the second half of BASELINE's metric, first-class next to the fit rate
            out["dcca_loss_fwd_bwd_per_s"] = extra["dcca_loss_metric_shape"]["value"]
            out["dcca_loss_roofline"] = extra["dcca_loss_metric_shape"]["roofline"]
        # last: its BLAS threads keep spinning for a while and would slow the launch chains of the GPU extras
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.n, a.d, a.k, a.cpu_sample_rows, a.cpu_runs)
            if not a.no_extras:
                out["cpu_baseline"]["c2_rcca"] = cpu_c2_whole()
                out["cpu_baseline"]["dcca_loss_configs3"] = cpu_loss_baseline()
            if not a.no_cpu_mcca and not a.no_extras:
                out["cpu_baseline"]["mcca_configs2"] = cpu_mcca_baseline()
                out["cpu_baseline"]["gcca_configs4"] = cpu_gcca_baseline()
        line = json.dumps(out)
    else:
        line = None
    if distributed:
        if comm is not None:
            comm.close()
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes its version banner into the C stdout buffer (it would be flushed at exit, AFTER the result).  The
    # contract is ONE JSON line on rank 0's stdout: drain the C buffer with fd 1 pointed at stderr, then print.
    sys.stdout.flush()
    try:
        import ctypes

        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    except Exception:
        pass
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
