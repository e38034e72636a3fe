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
    ap.add_argument("--route", default="auto", choices=["auto", "fp32", "bf16x2"], help="K1 arithmetic route (ccz_k1_route)")
    ap.add_argument("--fill", default="randn", choices=["randn", "zeros", "latent"], help="operand data (DVFS: zero-filled operands clock higher)")
    ap.add_argument("--offset", type=float, default=0.0, help="shift every column by this many standard deviations (pilot-shifted K1)")
    a = ap.parse_args()
    import torch

    from cca_zoo_amd import _backend

    h = _backend.default_handle(0)
    tdt = torch.float32 if a.dtype == "f32" else torch.float64
    es = 4 if a.dtype == "f32" else 8
    h.k1_route(a.route)
    if a.fill == "zeros":
        views = [torch.zeros(a.n, a.d, device="cuda", dtype=tdt) for _ in range(a.views)]
    elif a.fill == "latent":
        z = torch.randn(a.n, 64, device="cuda", dtype=tdt)
        views = []
        for _ in range(a.views):
            v = torch.randn(a.n, a.d, device="cuda", dtype=tdt)
            v.addmm_(z, torch.randn(64, a.d, device="cuda", dtype=tdt))
            views.append(v)
        del z
    else:
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
        r, sp, mf, rd = h.moments_last_route()
        if r == "bf16x2":
            print(f"iter {i}: route bf16x2: split {sp:.3f} ms, mfma {mf:.3f} ms = {3 * flop / mf / 1e9:.1f} TFLOP/s executed bf16 "
                  f"({flop / mf / 1e9:.1f} algorithmic), reduce {rd:.3f} ms", flush=True)
        print(f"iter {i}: gram {g:.3f} ms = {flop / g / 1e9:.2f} TFLOP/s ({a.n * D * es / g / 1e6:.1f} GB/s algorithmic), colsum {cs:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
