#!/usr/bin/env python3
"""ccz_transform on a large fp32 view: both arithmetic routes against a float64 product on the device, and their times.
    python tools/transform_probe.py [n] [d] [k]"""
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
torch.manual_seed(0)
h = _backend.default_handle(0)
X = torch.randn(n, d, device="cuda") * 1.5 + 0.75
mean = X[:65536].double().mean(0)
W = torch.randn(d, k, dtype=torch.float64, device="cuda") / d ** 0.5
ref = torch.empty(n, k, dtype=torch.float64, device="cuda")
for r0 in range(0, n, 65536):
    ref[r0:r0 + 65536] = (X[r0:r0 + 65536].double() - mean) @ W
out = torch.empty(n, k, device="cuda")
torch.cuda.synchronize()
for route in ("fp32", "bf16x2", "bf16x2", "fp32"):
    h.k1_route(route)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.check(h.lib.ccz_transform(h.raw, _backend.F32, C.c_void_p(X.data_ptr()), n, d, X.stride(0), C.c_void_p(mean.data_ptr()),
                                    C.c_void_p(W.data_ptr()), k, C.c_void_p(out.data_ptr()), out.stride(0)))
        h.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    rel = float((out.double() - ref).norm() / ref.norm())
    ms = min(ts)
    print(f"route {route}: {ms:.3f} ms (runs {[round(t, 3) for t in ts]}) = {n * d * 4 / ms / 1e6:.0f} GB/s algorithmic, "
          f"{2.0 * n * d * k / ms / 1e9:.1f} TFLOP/s; max err / max |ref| {err:.2e}, rel Frobenius {rel:.2e}", flush=True)
h.k1_route("auto")
