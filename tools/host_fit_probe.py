#!/usr/bin/env python3
"""PCIe-inclusive rate: rCCA.fit on pageable NumPy inputs (the reference's own calling convention)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend
from cca_zoo_amd.linear import rCCA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(0)
base = [rng.standard_normal((8192, d), dtype=np.float32) for _ in range(2)]
views = [np.tile(b, (n // 8192, 1)) for b in base]
gb = sum(v.nbytes for v in views) / 1e9
print(f"host views: n={n} d={d} fp32, {gb:.2f} GB pageable; cpu_count={os.cpu_count()}", flush=True)
h = _backend.default_handle(0)
D = 2 * d
mom = h.alloc((D * D + D) * 8)
for it in range(3):
    t0 = time.perf_counter()
    h.moments([(v, d, d) for v in views], n, _backend.F32, False, mom.ptr)
    h.sync()
    dt = time.perf_counter() - t0
    print(f"moments (host inputs): {dt*1e3:.1f} ms  {gb/dt:.1f} GB/s host->GPU inclusive  ({n*D*(D+1)/dt/1e12:.1f} TFLOP/s)", flush=True)
for it in range(2):
    t0 = time.perf_counter()
    m = rCCA(latent_dimensions=64, c=0.1).fit(views)
    dt = time.perf_counter() - t0
    print(f"rCCA.fit (host inputs): {dt*1e3:.1f} ms", flush=True)
dv = [torch.from_numpy(v).cuda() for v in views]
torch.cuda.synchronize()
for it in range(2):
    t0 = time.perf_counter()
    m = rCCA(latent_dimensions=64, c=0.1).fit(dv)
    h.sync()
    dt = time.perf_counter() - t0
    print(f"rCCA.fit (HBM-resident): {dt*1e3:.1f} ms", flush=True)
