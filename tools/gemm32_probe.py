#!/usr/bin/env python3
"""fp32 sample-side GEMM (loss backward / wide transform): (n x d) @ (d x k) through ccz_transform."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend
h = _backend.default_handle(0)
n, d, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1_000_000, 4096, 4096)
X = torch.randn(n, d, device="cuda")
W = torch.randn(d, k, device="cuda", dtype=torch.float64)
mean = torch.zeros(d, device="cuda", dtype=torch.float64)
out = torch.empty(n, k, device="cuda")
torch.cuda.synchronize()
def once():
    h.check(h.lib.ccz_transform(h.raw, _backend.F32, C.c_void_p(X.data_ptr()), n, d, d, C.c_void_p(mean.data_ptr()),
                                C.c_void_p(W.data_ptr()), k, C.c_void_p(out.data_ptr()), k))
once(); h.sync()
t0 = time.perf_counter()
for _ in range(3): once()
h.sync()
t = (time.perf_counter() - t0) / 3
ref = X[:512].double() @ W
err = float((out[:512].double() - ref).abs().max() / ref.abs().max())
print(f"n={n} d={d} k={k} impl={os.environ.get('CCZ_GEMM_NN_IMPL','1')}: {t*1e3:.1f} ms  {2.0*n*d*k/t/1e12:.1f} TFLOP/s  err {err:.1e}", flush=True)
