#!/usr/bin/env python3
"""Feasibility study of a split-bf16 K1 (SURVEY.md:288) -- a measurement TOOL, not a product path.

fp32 x = hi + mid + lo with three bf16 planes (8 + 8 + 8 mantissa bits); X'X is then the sum of the six plane products whose
terms are >= 2^-16 of the leading one (hi'hi, hi'mid + mid'hi, hi'lo + lo'hi, mid'mid), each on the bf16 MFMA pipe with
fp32 accumulation per row chunk and fp64 accumulation across chunks -- the same accumulation scheme as k_gram_f32_fifo.

What this script measures on one MI355X, with the VENDOR bf16 GEMM (hipBLASLt through torch.mm) standing in for a
hand-written kernel:
  * the error of the split Gram against float64 moments of the same fp32 data, next to the error of the fp32 product kernel
    (libccz K1) on the same rows -- the bar VERDICT set: no worse than the fp32 kernel's ~1.4e-6;
  * the rate the six products reach as A'B with the sample axis as K (the layout K1 has: X is row-major n x D, so both
    operands are "transposed"), and the cost of the split pass;
so that the decision "build the kernel / drop it" rests on numbers.  Usage: python tools/k1_split_probe.py [rows] [D]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cca_zoo_amd import _backend
from cca_zoo_amd._moments import compute_moments

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
chunk = 16384
torch.manual_seed(0)
dev = "cuda"
# data like the bench's: latent signal + noise, a mild mean
z = torch.randn(rows, 64, device=dev)
X = (z @ torch.randn(64, D, device=dev) + torch.randn(rows, D, device=dev)).contiguous()
del z


def split3(x):
    hi = x.to(torch.bfloat16)
    r1 = x - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def sync():
    torch.cuda.synchronize()


# ---- reference: float64 Gram (chunked) ----
G64 = torch.zeros(D, D, dtype=torch.float64, device=dev)
for r0 in range(0, rows, chunk):
    xc = X[r0:r0 + chunk].double()
    G64.addmm_(xc.T, xc)
    del xc
scale = torch.sqrt(torch.outer(torch.diag(G64), torch.diag(G64)))

# ---- split-bf16 Gram: six products per chunk, fp32 per chunk, fp64 across ----
pairs = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]


def split_gram(timed=False):
    G = torch.zeros(D, D, dtype=torch.float64, device=dev)
    t_split = t_mm = 0.0
    for r0 in range(0, rows, chunk):
        xc = X[r0:r0 + chunk]
        sync(); t0 = time.perf_counter()
        planes = split3(xc)
        sync(); t1 = time.perf_counter()
        acc = torch.zeros(D, D, dtype=torch.float32, device=dev)
        for a, b in pairs:
            acc += torch.mm(planes[a].T, planes[b], out_dtype=torch.float32)   # bf16 MFMA, fp32 accumulate AND fp32 result
        sync(); t2 = time.perf_counter()
        G += acc.double()
        t_split += t1 - t0
        t_mm += t2 - t1
    return G, t_split, t_mm


split_gram()                                                          # warm-up (hipBLASLt heuristics, allocator)
Gs, t_split, t_mm = split_gram()
err_split = float(((Gs - G64).abs() / scale).max())

# a single big bf16 product for the library's rate at this shape (K = rows of one chunk)
hi = X[:chunk].to(torch.bfloat16)
torch.mm(hi.T, hi, out_dtype=torch.float32); sync()
t0 = time.perf_counter()
for _ in range(5):
    torch.mm(hi.T, hi, out_dtype=torch.float32)
sync()
t_one = (time.perf_counter() - t0) / 5
bf16_tflops = 2.0 * chunk * D * D / t_one / 1e12

# ---- the product's fp32 K1 on the same rows ----
h = _backend.default_handle()
compute_moments([X[:, :D // 2], X[:, D // 2:]], h)                    # warm
mom, keep, nt, dims, kind = compute_moments([X[:, :D // 2], X[:, D // 2:]], h)
k1_ms = h.moments_last_ms()[0]
Gk = keep[0][: D * D].reshape(D, D)
err_k1 = float(((torch.triu(Gk) - torch.triu(G64)).abs() / scale).max())
flop = float(rows) * D * (D + 1)
out = {
    "rows": rows, "D": D, "chunk_rows": chunk,
    "fp32_kernel": {"ms": k1_ms, "tflops_algorithmic": flop / (k1_ms * 1e-3) / 1e12, "max_rel_err_vs_fp64": err_k1},
    "split_bf16x3_vendor_gemm": {"products": len(pairs), "split_pass_ms": t_split * 1e3, "gemm_ms": t_mm * 1e3,
                                 "total_ms": (t_split + t_mm) * 1e3, "max_rel_err_vs_fp64": err_split,
                                 "equivalent_tflops_algorithmic": flop / (t_split + t_mm) / 1e12,
                                 "note": "full D x D products (no symmetry), torch.mm = hipBLASLt, fp32 accumulate"},
    "vendor_bf16_gemm_rate_tflops_AtB": bf16_tflops,
}
print(json.dumps(out, indent=1))
