#!/usr/bin/env python3
"""CCALoss fwd+bwd at BASELINE configs[3] (batch 8192, 2 x 512, fp32) for rocprofv3 --kernel-trace: a few warm-up
calls, then ITERS timed calls; prints the per-call wall times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cca_zoo_amd.deep.objectives import CCALoss

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
torch.manual_seed(0)
z1 = torch.randn(n, d, device="cuda", requires_grad=True)
z2 = (0.5 * z1.detach() + torch.randn(n, d, device="cuda")).requires_grad_(True)
obj = CCALoss(eps=1e-6)
for _ in range(3):
    obj([z1, z2]).backward()
torch.cuda.synchronize()
ts = []
for _ in range(iters):
    z1.grad = None; z2.grad = None
    t0 = time.perf_counter()
    obj([z1, z2]).backward()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("ms per fwd+bwd:", " ".join(f"{t:.3f}" for t in ts), flush=True)
print("median", sorted(ts)[len(ts) // 2], flush=True)
