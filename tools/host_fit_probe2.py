import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cca_zoo_amd.linear import CCA
from cca_zoo_amd import _backend
rows, d = 262144, 4096
g = torch.Generator(device='cuda').manual_seed(0)
z = torch.randn(rows, 64, device='cuda', generator=g)
views = [(z @ torch.randn(64, d, device='cuda', generator=g) + torch.randn(rows, d, device='cuda', generator=g)) for _ in range(2)]
pin = [v.cpu().pin_memory() for v in views]
hv = [p.numpy() for p in pin]
del views, z
h = _backend.default_handle()
for _ in range(2): CCA(latent_dimensions=64).fit(hv)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); m = CCA(latent_dimensions=64).fit(hv); ts.append(time.perf_counter() - t0)
print('fit_s', [round(t, 4) for t in ts], 'timings', m.timings_)
