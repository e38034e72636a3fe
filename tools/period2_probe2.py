#!/usr/bin/env python3
"""bench.py's timed loop (model.fit on resident views, GC off) with ONE thing changed per variant: which change removes
the strict every-other-fit slow solve?  (tools/period2_probe.py's hand-written K1 -> solve loop does not show it.)

    plain      model.fit(views)                                   -- bench.py
    gc         automatic GC left on
    pause      host sleeps 3 ms between K1 and the solve (monkeypatched into _fit_moments)
    spin       host spins 3 ms (busy) between K1 and the solve
    fresh      a new CCA object per fit
    nostore    solve results are not stored on the model (no numpy conversions)
    manual     compute_moments + h.rcca_solve by hand, no estimator
    phases     plain under CCZ_TRACE_PHASES=1 (separate process: the switch is read once)
"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cca_zoo_amd import _backend
from cca_zoo_amd._moments import compute_moments
from cca_zoo_amd.datasets import JointData
from cca_zoo_amd.linear import CCA

variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["plain"]
fits = int(sys.argv[2]) if len(sys.argv) > 2 else 12
preheat = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
n, d, k = 1_000_000, 4096, 64
jd = JointData(n_views=2, n_samples=n, latent_dimensions=k, n_features=[d, d], random_state=0, latent_scales=list(np.linspace(2.0, 0.5, k)))
views = jd.sample_device(device="cuda:0", dtype=torch.float32, n_samples=n, seed=20260)
h = _backend.default_handle(0)
m = CCA(latent_dimensions=k)
t_end = time.perf_counter() + preheat
while time.perf_counter() < t_end:
    m.fit(views)
for _ in range(4):
    m.fit(views)
torch.cuda.synchronize()

orig_fit_moments = CCA._fit_moments
for variant in variants:
    model = CCA(latent_dimensions=k)
    CCA._fit_moments = orig_fit_moments
    if variant == "pause":
        def fm(self, *a, **kw):
            time.sleep(0.003)
            return orig_fit_moments(self, *a, **kw)
        CCA._fit_moments = fm
    elif variant == "spin":
        def fm(self, *a, **kw):
            t = time.perf_counter()
            while time.perf_counter() - t < 0.003:
                pass
            return orig_fit_moments(self, *a, **kw)
        CCA._fit_moments = fm
    elif variant == "nostore":
        def fm(self, hh, mom, n_total, dims, kind):
            hh.rcca_solve(mom, n_total, dims, [0.0, 0.0], self.center, self.latent_dimensions)
        CCA._fit_moments = fm
    gc.collect()
    if variant != "gc":
        gc.disable()
    rows = []
    for it in range(fits):
        t0 = time.perf_counter()
        if variant == "manual":
            mom, keep, nt, dims, kind = compute_moments(views, h)
            t1 = time.perf_counter()
            h.rcca_solve(mom, nt, dims, [0.0, 0.0], True, k)
            sv = (time.perf_counter() - t1) * 1e3
            del keep
        else:
            if variant == "fresh":
                model = CCA(latent_dimensions=k)
            model.fit(views)
            sv = model.timings_["solve_ms"]
        rows.append(((time.perf_counter() - t0) * 1e3, sv, h.moments_last_ms()[0]))
    gc.enable()
    s = np.array([r[1] for r in rows])
    print(f"## {variant}: solve series {np.round(s, 1).tolist()}", flush=True)
    print(f"## {variant}: fit mean {np.mean([r[0] for r in rows]):.1f} gram {np.mean([r[2] for r in rows]):.1f} solve min {s.min():.1f} mean {s.mean():.1f} "
          f"even {s[0::2].mean():.1f} odd {s[1::2].mean():.1f}", flush=True)
CCA._fit_moments = orig_fit_moments
