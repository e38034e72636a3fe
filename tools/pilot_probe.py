#!/usr/bin/env python3
"""K1 rate on fp32 data far from zero (every column mean = `offset` sigma) at the metric shape: the pilot shift inside the
FIFO kernel (default) against the register-staged pilot kernel (CCZ_GRAM_FIFO_PILOT=0) and against centred data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cca_zoo_amd import _backend
from cca_zoo_amd._moments import compute_moments

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = 4096
h = _backend.default_handle(0)
views = [torch.randn(n, d, device="cuda") for _ in range(2)]
flop = float(n) * 2 * d * (2 * d + 1)
for off in (0.0, 10.0):
    for v in views:
        v.add_(off)
    for it in range(3):
        mom, keep, _, _, _ = compute_moments(views, h)
        g, cs = h.moments_last_ms()
        print(f"offset {off:5.1f} it {it}: gram {g:8.2f} ms = {flop / g / 1e9:6.1f} TF  colsum {cs:.2f} ms  pilot {h.moments_last_pilot()}", flush=True)
        del keep
    for v in views:
        v.sub_(off)
