#!/usr/bin/env python3
"""Eight CCA fits at the metric shape (n reduced) with the pool / graph tracing on: which fits re-capture graphs or
miss the scratch pool?  (bench.py showed the solve alternating 27 / 52 ms.)"""
import os, sys, time
os.environ["CCZ_TRACE_POOL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd.datasets import JointData
from cca_zoo_amd.linear import CCA

n, d, k = int(os.environ.get("N", "1000000")), 4096, 64
jd = JointData(n_views=2, n_samples=n, latent_dimensions=k, n_features=[d, d], random_state=0, latent_scales=list(np.linspace(2.0, 0.5, k)))
views = jd.sample_device(device="cuda", dtype=torch.float32, n_samples=n, seed=1)
m = CCA(latent_dimensions=k)
for it in range(8):
    sys.stderr.write(f"---- fit {it}\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    m.fit(views)
    torch.cuda.synchronize()
    sys.stderr.write(f"fit {it}: {1e3 * (time.perf_counter() - t0):.1f} ms, solve {m.timings_['solve_ms']:.1f} ms\n")
