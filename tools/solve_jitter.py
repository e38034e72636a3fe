#!/usr/bin/env python3
"""Back-to-back rCCA solves on fixed moments: per-solve wall time and the GPU clock state, to see what the
solve chain (about 1100 dependent small kernels) is sensitive to."""
import glob, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend
from cca_zoo_amd._moments import compute_moments

def sclk():
    out = []
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            cur = [l.strip() for l in open(p) if "*" in l]
            out.append(cur[0] if cur else "?")
        except Exception as e:
            out.append(f"err {e}")
    return out[:2]

h = _backend.default_handle(0)
d, k, n = 4096, 64, 40000
torch.manual_seed(0)
z = torch.randn(n, k, device="cuda") * torch.linspace(2.0, 0.5, k, device="cuda")
views = [z @ torch.randn(k, d, device="cuda") + torch.randn(n, d, device="cuda") for _ in range(2)]
mom, keep, nt, dd, kd = compute_moments(views, h); h.sync()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
heavy = torch.randn(8192, 8192, device="cuda")
times = []
for it in range(iters):
    if len(sys.argv) > 2 and it % 25 == 24:          # a burst of heavy MFMA work between solves
        for _ in range(40): heavy @ heavy
        torch.cuda.synchronize()
    if os.environ.get('GAP_MS'): time.sleep(float(os.environ['GAP_MS']) * 1e-3)
    t0 = time.perf_counter()
    h.rcca_solve(mom, nt, dd, [0.1, 0.1], True, k)
    times.append((time.perf_counter() - t0) * 1e3)
    if it % 25 == 0: print(f"it {it}: {times[-1]:.1f} ms  sclk {sclk()}", flush=True)
t = np.array(times)
print("solve ms: min %.1f median %.1f mean %.1f p90 %.1f max %.1f" % (t.min(), np.median(t), t.mean(), np.percentile(t, 90), t.max()))
print("series:", np.round(t, 0).astype(int).tolist())
