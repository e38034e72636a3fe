#!/usr/bin/env python3
"""Micro-driver for K1 only (used under rocprofv3 / PMC collection).

    python tools/gram_probe.py --n 262144 --d 4096 --views 2 --dtype f32 --iters 3
"""

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=262144)
    ap.add_argument("--d", type=int, default=4096)
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--offset", type=float, default=0.0, help="shift every column by this many standard deviations (pilot-shifted K1)")
    a = ap.parse_args()
    import torch

    from cca_zoo_amd import _backend

    h = _backend.default_handle(0)
    tdt = torch.float32 if a.dtype == "f32" else torch.float64
    es = 4 if a.dtype == "f32" else 8
    views = [torch.randn(a.n, a.d, device="cuda", dtype=tdt) for _ in range(a.views)]
    if a.offset:
        for v in views:
            v.add_(a.offset)
    D = a.d * a.views
    mom = torch.empty(D * D + D, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    descr = [(v.data_ptr(), a.d, a.d) for v in views]
    flop = float(a.n) * D * (D + 1)
    for i in range(a.iters):
        h.moments(descr, a.n, _backend.F32 if a.dtype == "f32" else _backend.F64, True, mom.data_ptr())
        g, cs = h.moments_last_ms()
        print(f"iter {i}: gram {g:.3f} ms = {flop / g / 1e9:.2f} TFLOP/s ({a.n * D * es / g / 1e6:.1f} GB/s algorithmic), colsum {cs:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
