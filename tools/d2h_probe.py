#!/usr/bin/env python3
"""Where do the 16 / 25 ms of an outlier solve go?  ("no outliers".)  A measurement TOOL.

The outliers of bench.py's extras sit in the back-projection phase of the rCCA solve at the metric shape (three read-backs of
2 MB, 2 MB and 64 KB).  This script fits CCA on resident views n x (4096, 4096) a few times under each read-back mode of
ops_hip.hip's d2h() (CCZ_D2H_MODE) with CCZ_TRACE_D2H=1, which prints, per read-back, the time between two events around the
copy on the DEVICE time line and the host's wall time for the whole call.  Usage: python tools/d2h_probe.py [rows] [fits]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cca_zoo_amd import _backend
from cca_zoo_amd.datasets import JointData
from cca_zoo_amd.linear import CCA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
fits = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.environ["CCZ_TRACE_D2H"] = "1"
jd = JointData(n_views=2, n_samples=1, latent_dimensions=64, n_features=[4096, 4096], random_state=1, latent_scales=list(np.linspace(2.0, 0.5, 64)))
for tdt in (torch.float32, torch.float64):
    views = jd.sample_device(device="cuda", dtype=tdt, n_samples=n, seed=5)
    for mode in ("0", "1", "2", "3", "4", "0"):
        os.environ["CCZ_D2H_MODE"] = mode
        solves = []
        for i in range(fits):
            print(f"--- {tdt} mode {mode} fit {i}", file=sys.stderr, flush=True)
            m = CCA(latent_dimensions=64).fit(views)
            solves.append(round(m.timings_["solve_ms"], 2))
        print(f"{tdt} mode {mode}: solve ms {solves}", flush=True)
    del views
    torch.cuda.empty_cache()
