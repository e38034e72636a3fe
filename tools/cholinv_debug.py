#!/usr/bin/env python3
"""Where does the device factorization leave the NumPy one?  (ccz_cholinv fills L / X before it reports a bad pivot.)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cca_zoo_amd import _backend
H = _backend.default_handle(0)
rng = np.random.default_rng(0)
for d in [int(x) for x in (sys.argv[1:] or ["8", "16", "17", "32", "64"])]:
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    A = (q * np.geomspace(1.0, 1e-3, d)) @ q.T
    a, l, x = H.to_device(A), H.to_device(np.full((d, d), np.nan)), H.to_device(np.zeros((d, d)))
    arr = lambda p: (C.c_void_p * 1)(p)
    msg = "ok"
    try:
        H.check(H.lib.ccz_cholinv(H.raw, 1, arr(a.ptr), (C.c_int64 * 1)(d), arr(l.ptr), arr(x.ptr)))
    except Exception as e:
        msg = str(e)
    L = np.tril(np.nan_to_num(H.to_host(l, (d, d)), nan=7e77))
    X = np.tril(H.to_host(x, (d, d)))
    Lr = np.linalg.cholesky(A)
    err = np.abs(L - Lr)
    bad_cols = np.flatnonzero(err.max(axis=0) > 1e-9)
    print(f"d={d}: {msg}; max|L-Lref|={err.max():.3e}; first bad column {bad_cols[:1]}, bad rows in it "
          f"{np.flatnonzero(err[:, bad_cols[0]] > 1e-9)[:8] if len(bad_cols) else []}; |X L - I|={np.abs(X @ Lr - np.eye(d)).max():.3e}", flush=True)
    if len(bad_cols):
        c0 = bad_cols[0]
        print("   L[:, c] dev", L[c0:c0 + 6, c0], "ref", Lr[c0:c0 + 6, c0])
