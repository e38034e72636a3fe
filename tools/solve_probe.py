#!/usr/bin/env python3
"""Solve-stage timing for MCCA / GCCA / rCCA shapes (moments from a small n so the solve dominates)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend
from cca_zoo_amd._moments import compute_moments
h = _backend.default_handle(0)
kind = sys.argv[1]; dims = [int(x) for x in sys.argv[2].split(",")]; k = int(sys.argv[3]); n = int(sys.argv[4]) if len(sys.argv) > 4 else 40000
torch.manual_seed(0)
z = torch.randn(n, k, device="cuda") * torch.linspace(2.0, 0.5, k, device="cuda")
views = [z @ torch.randn(k, d, device="cuda") + torch.randn(n, d, device="cuda") for d in dims]
for it in range(3):
    mom, keep, nt, dd, kd = compute_moments(views, h); h.sync()
    t0 = time.perf_counter()
    if kind == "rcca": W, mu, vals = h.rcca_solve(mom, nt, dd, [0.1, 0.1], True, k)
    elif kind == "mcca": W, mu, vals = h.mcca_solve(mom, nt, dd, [0.1] * len(dd), 1e-6, True, k)
    else: W, mu, vals = h.gcca_solve(mom, nt, dd, [0.1] * len(dd), [1.0] * len(dd), 1e-6, True, k)
    print(f"{kind} dims={dims} k={k}: solve {1e3*(time.perf_counter()-t0):.1f} ms  vals[:3]={vals[:3]}", flush=True)
