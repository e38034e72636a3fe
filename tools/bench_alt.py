#!/usr/bin/env python3
"""bench.py's timed loop, instrumented: does the solve alternate between ~19 and ~44 ms, and what correlates with it?
Variants (VARIANT env): plain | sync (torch.cuda.synchronize() after every fit) | gc (automatic GC left on)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend, _moments
from cca_zoo_amd.datasets import JointData
from cca_zoo_amd.linear import CCA

variant = os.environ.get("VARIANT", "plain")
n, d, k = 1_000_000, 4096, 64
jd = JointData(n_views=2, n_samples=n, latent_dimensions=k, n_features=[d, d], random_state=0, latent_scales=list(np.linspace(2.0, 0.5, k)))
views = jd.sample_device(device="cuda:0", dtype=torch.float32, n_samples=n, seed=20260)
h = _backend.default_handle(0)
m = CCA(latent_dimensions=k)
for _ in range(5):
    m.fit(views)
torch.cuda.synchronize()
gc.collect()
if variant != "gc":
    gc.disable()
rows = []
for it in range(10):
    t0 = time.perf_counter()
    m.fit(views)
    if variant == "sync":
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    rows.append((1e3 * (t1 - t0), m.timings_["moments_ms"], m.timings_["solve_ms"], h.moments_last_ms()[0]))
for r in rows:
    print(f"{variant}: fit {r[0]:.1f} ms  moments(wall) {r[1]:.1f}  solve {r[2]:.1f}  gram(events) {r[3]:.1f}", flush=True)
