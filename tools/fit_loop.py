#!/usr/bin/env python3
"""Per-fit wall time distribution at the headline shape (spots periodic stalls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend
from cca_zoo_amd.datasets import JointData
from cca_zoo_amd.linear import CCA
from cca_zoo_amd._moments import compute_moments

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, k = 4096, 64
h = _backend.default_handle(0)
jd = JointData(n_views=2, n_samples=n, latent_dimensions=k, n_features=[d, d], random_state=0,
               latent_scales=list(np.linspace(2.0, 0.5, k)))
views = jd.sample_device(device="cuda:0", dtype=torch.float32, n_samples=n, seed=1)
torch.cuda.synchronize()
m = CCA(latent_dimensions=k)
for it in range(12):
    t0 = time.perf_counter()
    mom, keep, nt, dims, kind = compute_moments(views, h)
    h.sync()
    t1 = time.perf_counter()
    W, means, vals = h.rcca_solve(mom, nt, dims, [0.0, 0.0], True, k)
    t2 = time.perf_counter()
    del keep, mom
    t3 = time.perf_counter()
    print(f"it {it}: moments {1e3*(t1-t0):.1f} (gram {h.moments_last_ms()[0]:.1f}) solve {1e3*(t2-t1):.1f} free {1e3*(t3-t2):.2f}", flush=True)
