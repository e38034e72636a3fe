#!/usr/bin/env python3
"""Time ccz_cca_loss forward+backward at a given shape and check it against the fp64 closed form on a row subset."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd.deep.objectives import CCALoss

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.manual_seed(0)
z1 = torch.randn(n, d, device="cuda")
z2 = (0.5 * z1 + torch.randn(n, d, device="cuda"))
z1.requires_grad_(True); z2.requires_grad_(True)
obj = CCALoss(eps=1e-6)
for it in range(iters):
    z1.grad = None; z2.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = obj([z1, z2]); loss.backward()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    flop = 2.0 * n * d * d * (3 + 4)
    print(f"it {it}: fwd+bwd {dt*1e3:.2f} ms  ({flop/dt/1e12:.1f} TFLOP/s nominal)  loss {loss.item():.6f}", flush=True)
if n <= 20000:
    from oracle import losses as ol
    l, g1, g2 = ol.cca_loss_closed_form(z1.detach().cpu().numpy(), z2.detach().cpu().numpy(), 1e-6)
    e1 = np.linalg.norm(z1.grad.cpu().numpy() - g1) / np.linalg.norm(g1)
    print("loss rel err", abs(loss.item() - l) / abs(l), "grad rel err", e1)
