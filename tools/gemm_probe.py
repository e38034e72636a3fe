#!/usr/bin/env python3
"""Throughput of ccz_gemm_f64 on solver-like shapes."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cca_zoo_amd import _backend
h = _backend.default_handle(0)
def run(tA, tB, M, N, K, reps=5):
    A = torch.randn((K, M) if tA else (M, K), dtype=torch.float64, device="cuda")
    B = torch.randn((N, K) if tB else (K, N), dtype=torch.float64, device="cuda")
    Cm = torch.zeros((M, N), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    def once():
        h.check(h.lib.ccz_gemm_f64(h.raw, tA, tB, M, N, K, 1.0, C.c_void_p(A.data_ptr()), A.shape[1], C.c_void_p(B.data_ptr()), B.shape[1], 0.0, C.c_void_p(Cm.data_ptr()), N))
    once(); h.sync()
    t0 = time.perf_counter()
    for _ in range(reps): once()
    h.sync()
    dt = (time.perf_counter() - t0) / reps
    ref = (A.T if tA else A) @ (B.T if tB else B)
    err = float((Cm - ref).abs().max() / ref.abs().max())
    print(f"tA={tA} tB={tB} M={M} N={N} K={K}: {dt*1e6:.1f} us  {2.0*M*N*K/dt/1e12:.1f} TFLOP/s  err {err:.1e}", flush=True)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for shp in shapes or [(0,1,2048,2048,2048),(0,0,2048,2048,2048),(1,0,2048,2048,2048),(1,1,2048,2048,2048),(0,1,4096,2048,2048),(0,1,4032,4032,64),(0,1,1024,1024,1024),(0,0,4096,80,4096),(1,0,4096,80,4096),(0,1,8192,8192,4096)]:
    run(*shp)
