#!/usr/bin/env python3
"""The fp32 projection kernel (ccz_transform, k <= 64) in both forms -- a row per lane (CCZ_TALL_IMPL=1) and whole-line loads
(2) -- on one view: time by HIP events over back-to-back calls, error against a float64 product.
    python tools/tall_probe.py [n] [d] [k] [check]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cca_zoo_amd import _backend

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
check = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
torch.manual_seed(0)
h = _backend.default_handle(0)
h.k1_route("fp32")
X = torch.randn(n, d, device="cuda") * 1.5 + 0.75
mean = X[:65536].double().mean(0)
W = torch.randn(d, k, dtype=torch.float64, device="cuda") / d ** 0.5
out = torch.empty(n, k, device="cuda")
ref = None
if check:
    ref = torch.empty(n, k, dtype=torch.float64, device="cuda")
    for r0 in range(0, n, 65536):
        ref[r0:r0 + 65536] = (X[r0:r0 + 65536].double() - mean) @ W


def call():
    h.check(h.lib.ccz_transform(h.raw, _backend.F32, C.c_void_p(X.data_ptr()), n, d, X.stride(0), C.c_void_p(mean.data_ptr()),
                                C.c_void_p(W.data_ptr()), k, C.c_void_p(out.data_ptr()), out.stride(0)))


for impl in (os.environ.get("TALL_IMPLS", "1,2,2,1")).split(","):
    os.environ["CCZ_TALL_IMPL"] = impl.rstrip("w")
    os.environ["CCZ_TALL_NJ1"] = "0" if impl.endswith("w") else "1"      # "3w": two column tiles also for k <= 32
    out.zero_()
    call()
    h.sync()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        h.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    ms = min(ts)
    msg = f"impl {impl}: n {n} d {d} k {k}: {ms:.3f} ms (runs {[round(t, 3) for t in ts]}) = {n * d * 4 / ms / 1e6:.0f} GB/s, {2.0 * n * d * k / ms / 1e9:.1f} TF"
    if ref is not None:
        msg += f"; max err / max |ref| {float((out.double() - ref).abs().max() / ref.abs().max()):.2e}"
    print(msg, flush=True)
