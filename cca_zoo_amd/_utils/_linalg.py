"""Device-backed drop-ins for the reference's two numerical seams.

* :func:`svd_whiten`  -- cca_zoo/_utils/_linalg.py:9-41.  The reference SVDs the
  n x d data; here the Gram X'X is assembled on the MFMA pipe (K1) and its
  eigendecomposition (one-sided Jacobi on the device) yields the same
  ``W = V ((1-c) lam + c)^-1/2`` up to column signs; ``X_white = X @ W``.
* :func:`gevp`        -- cca_zoo/_utils/_linalg.py:44-73.  Top-k symmetric
  (generalised) eigenpairs, descending, ``v' B v = 1``.

Inputs/outputs are NumPy arrays like the reference; all arithmetic runs in libccz.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from cca_zoo_amd import _backend


def svd_whiten(X: np.ndarray, regularization: float = 0.0) -> tuple[np.ndarray, np.ndarray]:
    X = np.ascontiguousarray(X)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    n, d = X.shape
    h = _backend.default_handle()
    code = _backend.F32 if X.dtype == np.float32 else _backend.F64
    mom = h.alloc((d * d + d) * 8)
    # X crosses the host link ONCE when it fits next to its whitened copy (it used to be streamed for the
    # moments and pushed a second time for X W): resident, K1 and the projection both read it from HBM
    resident = 2 * X.nbytes + (1 << 30) < 0.5 * float(h.device_info()["hbm_bytes"])
    Xd = h.to_device(X) if resident else None
    if resident:
        h.moments([(Xd.ptr, d, d)], n, code, True, mom.ptr)
    else:
        h.moments([(X, d, d)], n, code, False, mom.ptr)
    h.moments_symmetrize(mom.ptr, d)
    Wd = h.alloc(d * d * 8)
    lam = h.alloc(d * 8)
    r = C.c_int64(0)
    h.check(h.lib.ccz_whitener(h.raw, C.c_void_p(mom.ptr), d, n, float(regularization),
                               C.c_void_p(Wd.ptr), C.c_void_p(lam.ptr), C.byref(r)))
    W = h.to_host(Wd, (d, r.value)).astype(X.dtype, copy=False)
    # X_white = X W on the device
    out = h.alloc(n * r.value * X.itemsize)
    if Xd is None:
        Xd = h.to_device(X)
    Wd64 = h.to_device(np.ascontiguousarray(W, dtype=np.float64))
    h.check(h.lib.ccz_transform(h.raw, code,
                                C.c_void_p(Xd.ptr), n, d, d, None, C.c_void_p(Wd64.ptr), r.value,
                                C.c_void_p(out.ptr), r.value))
    X_white = h.to_host(out, (n, r.value), dtype=X.dtype)
    return X_white, W


def gevp(A: np.ndarray, B: np.ndarray | None, k: int) -> tuple[np.ndarray, np.ndarray]:
    A = np.ascontiguousarray(A, dtype=np.float64)
    p = A.shape[0]
    kk = min(int(k), p)
    h = _backend.default_handle()
    Ad = h.to_device(A)
    Bd = h.to_device(np.ascontiguousarray(B, dtype=np.float64)) if B is not None else None
    w = h.alloc(kk * 8)
    V = h.alloc(p * kk * 8)
    h.check(h.lib.ccz_gevp_topk(h.raw, C.c_void_p(Ad.ptr), C.c_void_p(Bd.ptr) if Bd is not None else None,
                                p, kk, C.c_void_p(w.ptr), C.c_void_p(V.ptr)))
    return h.to_host(w, (kk,)), h.to_host(V, (p, kk))
