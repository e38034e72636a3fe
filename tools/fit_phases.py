#!/usr/bin/env python3
"""Time the phases of one CCA fit at the headline shape (moments / solve), several repeats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend
from cca_zoo_amd._moments import compute_moments
from cca_zoo_amd.datasets import JointData

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, k = 4096, 64
h = _backend.default_handle(0)
jd = JointData(n_views=2, n_samples=n, latent_dimensions=k, n_features=[d, d], random_state=0,
               latent_scales=list(np.linspace(2.0, 0.5, k)))
views = jd.sample_device(device="cuda:0", dtype=torch.float32, n_samples=n, seed=1)
torch.cuda.synchronize()
for it in range(int(os.environ.get('PHASE_ITERS', '4'))):
    t0 = time.perf_counter()
    mom, keep, nt, dims, kind = compute_moments(views, h)
    h.sync()
    t1 = time.perf_counter()
    W, means, vals = h.rcca_solve(mom, nt, dims, [0.0, 0.0], True, k)
    t2 = time.perf_counter()
    print(f"it {it}: moments {1e3*(t1-t0):.1f} ms (gram kernel {h.moments_last_ms()[0]:.1f}, colsum {h.moments_last_ms()[1]:.1f}), solve {1e3*(t2-t1):.1f} ms, top corr {vals[:3]}", flush=True)

if os.environ.get('PHASE_ONLY'): sys.exit(0)
from cca_zoo_amd.linear import CCA
import cProfile, pstats
m = CCA(latent_dimensions=k)
for it in range(3):
    t0 = time.perf_counter(); m.fit(views); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"fit {it}: {1e3*(t1-t0):.1f} ms", flush=True)
pr = cProfile.Profile(); pr.enable(); m.fit(views); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
