#!/usr/bin/env python3
"""Gram-reuse grid search at the metric shape: GridSearchCV(rCCA, c grid, 5-fold) on HBM-resident views."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend
from cca_zoo_amd.datasets import JointData
from cca_zoo_amd.linear import rCCA
from cca_zoo_amd.model_selection import GridSearchCV

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
k = 64
jd = JointData(n_views=2, n_samples=n, latent_dimensions=k, n_features=[d, d], signal_to_noise=1.0, random_state=0,
               latent_scales=list(np.linspace(2.0, 0.5, k)))
views = jd.sample_device(device="cuda:0", dtype=torch.float32, n_samples=n, seed=1000)
torch.cuda.synchronize()
h = _backend.default_handle(0)
rCCA(latent_dimensions=k, c=0.1).fit(views); h.sync()
t0 = time.perf_counter(); rCCA(latent_dimensions=k, c=0.1).fit(views); h.sync(); t_fit = time.perf_counter() - t0
grid = {"c": [float(x) for x in os.environ["GRID_C"].split(",")]} if os.environ.get("GRID_C") else {"c": [1e-4, 1e-3, 1e-2, 0.05, 0.1, 0.3, 0.6, 0.9]}
for it in range(int(os.environ.get('GRID_ITERS', '2'))):
    t0 = time.perf_counter()
    gs = GridSearchCV(rCCA(latent_dimensions=k), grid, cv=5).fit(views)
    h.sync()
    dt = time.perf_counter() - t0
    nfit = len(grid["c"]) * 5 + 1
    print(f"GridSearchCV 8 settings x 5 folds + refit, n={n} 2x{d} fp32: {dt*1e3:.0f} ms "
          f"(moments pass {gs.moments_pass_time_*1e3:.0f} ms, mean solve {np.mean(gs.cv_results_['mean_fit_time'])*1e3:.1f} ms, "
          f"mean score {np.mean(gs.cv_results_['mean_score_time'])*1e3:.1f} ms); one plain fit {t_fit*1e3:.0f} ms -> "
          f"refit-per-setting equivalent {nfit} x {t_fit*1e3:.0f} = {nfit*t_fit:.1f} s; speed-up {nfit*t_fit/dt:.1f}x", flush=True)
    if os.environ.get("GRID_VERBOSE"):
        print("   per-setting mean fit ms:", np.round(gs.cv_results_["mean_fit_time"] * 1e3, 1), "refit", round(gs.refit_time_ * 1e3, 1))
print("best", gs.best_params_, "score", gs.best_score_, "mean_test_score", np.round(gs.cv_results_["mean_test_score"], 4))
