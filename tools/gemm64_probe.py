#!/usr/bin/env python3
"""The solver's fp64 GEMM (ccz_gemm_f64: k_gemm_f64_big / k_gemm_f64_half) shape by shape against the 78.6 TF fp64 matrix
peak and against torch.matmul (rocBLAS) on the same operands -- a measurement TOOL for DESIGN section 7 item 1.

  python tools/gemm64_probe.py            the shapes of the rCCA / GCCA solves (A B' with K = 512) + a K sweep
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cca_zoo_amd import _backend

h = _backend.default_handle() if hasattr(_backend, "default_handle") else _backend.Handle(0)
PEAK = 78.6e12


def timed(fn, iters):
    import time
    fn()
    torch.cuda.synchronize()
    h.sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    h.sync()
    return (time.perf_counter() - t0) / iters * 1e6   # us


def one(M, N, K, tA=False, tB=True, iters=20):
    A = torch.randn((K, M) if tA else (M, K), dtype=torch.float64, device="cuda")
    B = torch.randn((N, K) if tB else (K, N), dtype=torch.float64, device="cuda")
    Cc = torch.zeros(M, N, dtype=torch.float64, device="cuda")
    us = timed(lambda: h.gemm(tA, tB, M, N, K, 1.0, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), 0.0, Cc.data_ptr(), N), iters)
    opA = A.T if tA else A
    opB = B.T if tB else B
    ref = opA @ opB
    err = float((Cc - ref).abs().max() / ref.abs().max())
    out = torch.empty_like(ref)
    us_t = timed(lambda: torch.matmul(opA, opB, out=out), iters)
    F = 2.0 * M * N * K
    rec = {"M": M, "N": N, "K": K, "tA": int(tA), "tB": int(tB), "us": round(us, 1), "TF": round(F / us / 1e6, 1), "frac": round(F / us / 1e6 / 78.6, 3),
           "torch_us": round(us_t, 1), "torch_TF": round(F / us_t / 1e6, 1), "err": err}
    print(json.dumps(rec), flush=True)


def skinny():
    """tall-times-skinny products of the subspace iteration (k_gemm_f64_skinny / _skinny2): S X and S' X at the solver's shapes"""
    for M, N, K in [(4096, 80, 4096), (8192, 80, 8192), (16384, 160, 16384), (4096, 64, 4096), (4096, 96, 4096), (4096, 128, 4096),
                    (5000, 80, 4096), (4096, 160, 2048), (4098, 72, 1024), (16384, 176, 16384)]:
        for tA in (False, True):
            one(M, N, K, tA, False, iters=10)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "skinny":
        skinny()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "one":          # one M N K tA tB   (for a counter pass over ONE shape)
        one(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), bool(int(sys.argv[5])), bool(int(sys.argv[6])), iters=3)
        sys.exit(0)
    # ragged edges, all four transposition pairs, one and three k-blocks (correctness of the tile loop's ends)
    for tA in (False, True):
        for tB in (False, True):
            for M, N, K in [(1111, 999, 16), (1111, 999, 48), (1400, 1290, 144), (4001, 515, 512), (130, 4097, 64)]:
                one(M, N, K, tA, tB, iters=3)
    # triangular-solve / Cholesky-update shapes of the 2 x 4096 rCCA solve and the D = 16384 GCCA solve
    for M, N, K in [(4096, 512, 512), (4096, 3584, 512), (4096, 2048, 512), (4096, 1024, 512), (3584, 3584, 512), (2048, 2048, 512),
                    (16384, 512, 512), (16384, 8192, 512), (16384, 15872, 512), (4096, 4096, 4096)]:
        one(M, N, K)
    # K sweep on one full wave of tiles (256 tiles: one workgroup per CU) and on four per CU
    for K in (128, 256, 512, 1024, 2048):
        one(2048, 2048, K)
    for K in (128, 256, 512, 1024, 2048):
        one(4096, 4096, K)
    for tA, tB in ((False, False), (True, False)):
        one(4096, 3584, 512, tA, tB)
