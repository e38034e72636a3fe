#!/usr/bin/env python3
"""CCA fits at the metric shape on data at `offset` sigma (pilot-shifted K1): per-fit solve time (with CCZ_TRACE_PHASES=2 the
per-phase device / host times and shader clock)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend
from cca_zoo_amd.datasets import JointData
from cca_zoo_amd.linear import CCA

off = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
n, d, k = 1_000_000, 4096, 64
jd = JointData(n_views=2, n_samples=n, latent_dimensions=k, n_features=[d, d], random_state=1, latent_scales=list(np.linspace(2.0, 0.5, k)))
views = jd.sample_device(device="cuda:0", dtype=torch.float32, n_samples=n, seed=20261)
if off:
    for v in views:
        v.add_(off * float(v[:4096].std()))
h = _backend.default_handle(0)
gc.collect(); gc.disable()
for it in range(7):
    t0 = time.perf_counter()
    m = CCA(latent_dimensions=k).fit(views)
    torch.cuda.synchronize()
    print(f"offset {off}: fit {1e3 * (time.perf_counter() - t0):.1f} ms  gram {h.moments_last_ms()[0]:.1f}  solve {m.timings_['solve_ms']:.1f}  pilot {h.moments_last_pilot()}", flush=True)
