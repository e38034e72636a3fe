#!/usr/bin/env python3
"""One DCCA training step (two MLP encoders 784-1024-1024-512, batch 8192, CCALoss) in a loop, for rocprofv3 --kernel-trace:
wall time per step next to the traced kernel times (is the step GPU-bound or launch-bound?).  python tools/train_step_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from cca_zoo_amd.deep.objectives import CCALoss

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(0)
def mlp():
    return nn.Sequential(nn.Linear(784, 1024), nn.ReLU(), nn.Linear(1024, 1024), nn.ReLU(), nn.Linear(1024, 512)).cuda()
e1, e2 = mlp(), mlp()
x1 = torch.randn(8192, 784, device="cuda"); x2 = torch.randn(8192, 784, device="cuda")
obj = CCALoss(eps=1e-4)
params = list(e1.parameters()) + list(e2.parameters())
def step():
    for p in params:
        p.grad = None
    loss = obj([e1(x1), e2(x2)])
    loss.backward()
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"ms per step: {(time.perf_counter() - t0) / steps * 1e3:.4f}", flush=True)
