#!/usr/bin/env python3
"""Per-fit wall time and phase split of the first fits of a process (warm-up effects), at the metric shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend, _moments
from cca_zoo_amd.datasets import JointData
from cca_zoo_amd.linear import CCA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d, k = 4096, 64
h = _backend.default_handle(0)
jd = JointData(n_views=2, n_samples=n, latent_dimensions=k, n_features=[d, d], signal_to_noise=1.0, random_state=0,
               latent_scales=list(np.linspace(2.0, 0.5, k)))
views = jd.sample_device(device="cuda:0", dtype=torch.float32, n_samples=n, seed=1000)
torch.cuda.synchronize()
orig_cm, orig_solve = _moments.compute_moments, h.rcca_solve
marks = {}
def cm(*a, **kw):
    t = time.perf_counter(); r = orig_cm(*a, **kw); h.sync(); marks["moments"] = time.perf_counter() - t; return r
def sv(*a, **kw):
    t = time.perf_counter(); r = orig_solve(*a, **kw); marks["solve"] = time.perf_counter() - t; return r
import cca_zoo_amd.linear._rcca as R
R.compute_moments = cm
h.rcca_solve = sv
m = CCA(latent_dimensions=k)
for it in range(7):
    t0 = time.perf_counter(); m.fit(views); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"fit {it}: {1e3*(t1-t0):.1f} ms  moments {1e3*marks['moments']:.1f} (kernel {h.moments_last_ms()[0]:.1f}) solve {1e3*marks['solve']:.1f} "
          f"other {1e3*(t1-t0-marks['moments']-marks['solve']):.1f}", flush=True)
