#!/usr/bin/env python3
"""HBM roofline check of transform / score on HBM-resident views."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend
from cca_zoo_amd.linear import rCCA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dt = torch.float32 if (len(sys.argv) <= 4 or sys.argv[4] == "f32") else torch.float64
h = _backend.default_handle(0)
torch.manual_seed(0)
views = [torch.randn(n, d, device="cuda", dtype=dt) for _ in range(2)]
m = rCCA(latent_dimensions=k, c=0.1).fit([v[:65536] for v in views])
for name, fn in (("transform", lambda: m.transform(views)), ("score", lambda: m.score(views))):
    fn(); h.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    h.sync(); torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    gb = 2 * n * d * views[0].element_size() / 1e9
    print(f"{name}: {t*1e3:.1f} ms for 2 views n={n} d={d} k={k} {dt}: {gb/t:.0f} GB/s of input ({gb/t/8000*100:.0f}% of 8 TB/s), "
          f"{2*2.0*n*d*k/t/1e12:.1f} TFLOP/s", flush=True)
