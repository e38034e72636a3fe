#!/usr/bin/env python3
"""K1 routes side by side on one MI355X: the fp32 MFMA kernel and the split-bf16 route (ccz_k1_route) against float64
moments of the same fp32 rows -- a measurement TOOL (the gate of VERDICT r5 item 1), nothing here is on a product path.

  python tools/k1_route_check.py small          ragged / unaligned / shifted shapes: correctness of both routes
  python tools/k1_route_check.py big [n] [d]    two views n x d (default 1e6 x 4096): error and time of both routes
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cca_zoo_amd import _backend

dev = "cuda"
h = _backend.default_handle() if hasattr(_backend, "default_handle") else _backend.Handle(0)


def moments(views, route, timed=True, pilot=None):
    n = views[0].shape[0]
    D = sum(v.shape[1] for v in views)
    mom = torch.empty(D * D + D, dtype=torch.float64, device=dev)
    h.k1_route(route)
    vv = [(v.data_ptr(), v.shape[1], v.stride(0)) for v in views]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.moments(vv, n, _backend.F32 if hasattr(_backend, "F32") else 0, True, mom.data_ptr(), accumulate=False, pilot=pilot, timed=timed)
    h.sync()
    t1 = time.perf_counter()
    h.k1_route("auto")
    return mom[:D * D].view(D, D), mom[D * D:], (t1 - t0) * 1e3


def ref64(views, chunk=16384):
    n = views[0].shape[0]
    D = sum(v.shape[1] for v in views)
    G = torch.zeros(D, D, dtype=torch.float64, device=dev)
    s = torch.zeros(D, dtype=torch.float64, device=dev)
    for r0 in range(0, n, chunk):
        xc = torch.cat([v[r0:r0 + chunk].double() for v in views], dim=1)
        G.addmm_(xc.T, xc)
        s += xc.sum(0)
        del xc
    return G, s


def errs(G, s, G64, s64, n):
    """max relative error of the CENTRED second moments (what every solve reads), scaled by sqrt(C_ii C_jj); upper triangle."""
    C64 = G64 - torch.outer(s64, s64) / n
    C = G - torch.outer(s, s) / n
    sc = torch.sqrt(torch.outer(torch.diag(C64), torch.diag(C64)))
    iu = torch.triu(torch.ones_like(C64, dtype=torch.bool))
    e = ((C - C64).abs() / sc)[iu]
    rawsc = torch.sqrt(torch.outer(torch.diag(G64), torch.diag(G64)))
    er = ((G - G64).abs() / rawsc)[iu]
    return float(e.max()), float(e.pow(2).mean().sqrt()), float(er.max()), float((s - s64).abs().max() / s64.abs().max().clamp_min(1e-300))


def small():
    torch.manual_seed(1)
    out = []
    cases = [
        ("aligned 2x512 n=8192", [512, 512], 8192, 0.0, False),
        ("ragged 300+520 n=5000", [300, 520], 5000, 0.0, False),
        ("odd 257+63 n=4099 unaligned", [257, 63], 4099, 0.0, True),
        ("shifted mean=10 sigma 2x256 n=20000", [256, 256], 20000, 10.0, False),
        ("three views 128+384+200 n=33000", [128, 384, 200], 33000, 0.5, False),
        ("one k-step n=16", [256, 256], 16, 0.0, False),
        ("n=17", [256], 17, 0.0, False),
    ]
    for name, dims, n, shift, unaligned in cases:
        z = torch.randn(n, 8, device=dev)
        views = []
        for d in dims:
            x = z @ torch.randn(8, d, device=dev) + torch.randn(n, d, device=dev) + shift
            if unaligned:
                buf = torch.empty(n, d + 3, device=dev)
                buf[:, 1:d + 1] = x
                x = buf[:, 1:d + 1]
            views.append(x)
        G64, s64 = ref64(views)
        rec = {"case": name}
        for route in ("fp32", "bf16x2"):
            G, s, _ = moments(views, route)
            rec[route] = dict(zip(("cov_max", "cov_rms", "raw_max", "colsum"), errs(G, s, G64, s64, n)))
            rec[route]["route_taken"] = h.moments_last_route()[0]
        out.append(rec)
        print(json.dumps(rec), flush=True)
    bad = [r for r in out if r["bf16x2"]["cov_max"] > 2e-5 or r["bf16x2"]["route_taken"] != "bf16x2"]
    print("SMALL:", "FAIL" if bad else "ok")
    return 1 if bad else 0


def big(n, d):
    torch.manual_seed(0)
    k = 64
    z = torch.randn(n, k, device=dev)
    views = []
    for _ in range(2):
        x = torch.empty(n, d, device=dev)
        W = torch.randn(k, d, device=dev)
        for r0 in range(0, n, 65536):
            x[r0:r0 + 65536] = z[r0:r0 + 65536] @ W + torch.randn(min(65536, n - r0), d, device=dev)
        views.append(x)
    del z
    rec = {"n": n, "d": d}
    G64 = s64 = None
    if n * d <= 131072 * 4096 * 2:
        G64, s64 = ref64(views)
    F = float(n) * (2 * d) * (2 * d + 1)
    for route in ("fp32", "bf16x2", "bf16x2", "fp32", "bf16x2"):
        G, s, wall = moments(views, route)
        g_ms, cs_ms = h.moments_last_ms()
        r, sp, mf, rd = h.moments_last_route()
        e = dict(zip(("cov_max", "cov_rms", "raw_max", "colsum"), errs(G, s, G64, s64, n))) if G64 is not None else {}
        row = {"route": r, "wall_ms": round(wall, 2), "gram_ms": round(g_ms, 2), "colsum_ms": round(cs_ms, 2), "split_ms": round(sp, 2),
               "mfma_ms": round(mf, 2), "reduce_ms": round(rd, 2), "alg_TF": round(F / (g_ms * 1e-3) / 1e12, 1),
               "exec_bf16_TF": round(3 * F / (mf * 1e-3) / 1e12, 1) if mf > 0 else None, **e}
        print(json.dumps(row), flush=True)
    return 0


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "small"
    if mode == "small":
        sys.exit(small())
    n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    sys.exit(big(n, d))
