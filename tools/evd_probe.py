#!/usr/bin/env python3
"""Warm timings of the dense seams behind ccz_syevj / ccz_gesvj (csrc/evd_block.hip).

    python tools/evd_probe.py syev 512,1024,4096 [reps]      python tools/evd_probe.py gesv 1024x1024,4096x1024 [reps]

Prints one line per size: sweeps, best / median ms, nominal rate (9 d^3 for the EVD, 21 p q min(p, q) ... for the SVD),
and the accuracy against LAPACK (numpy) when CCZ_PROBE_CHECK=1.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cca_zoo_amd import _backend  # noqa: E402


def main():
    what = sys.argv[1]
    sizes = sys.argv[2].split(",")
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    check = os.environ.get("CCZ_PROBE_CHECK", "0") == "1"
    H = _backend.default_handle(0)
    for sz in sizes:
        rng = np.random.default_rng(11)
        if what == "syev":
            d = int(sz)
            X = rng.standard_normal((3 * d, d)) * np.linspace(2.0, 0.05, d)
            A = X.T @ X / (3 * d - 1)
            Ad, wd, Vd = H.to_device(A), H.alloc(d * 8), H.alloc(d * d * 8)
            sw = C.c_int(0)
            ts = []
            for _ in range(reps + 1):
                H.sync()
                t0 = time.perf_counter()
                H.check(H.lib.ccz_syevj(H.raw, C.c_void_p(Ad.ptr), d, C.c_void_p(wd.ptr), C.c_void_p(Vd.ptr), C.byref(sw)))
                H.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            warm = sorted(ts[1:])
            line = (f"syev d={d}: sweeps {sw.value} first {ts[0]:.2f} ms best {warm[0]:.2f} ms median {warm[len(warm) // 2]:.2f} ms "
                    f"nominal 9d^3 rate {9 * d ** 3 / warm[0] / 1e9:.2f} TF")
            if check:
                w, V = H.to_host(wd, (d,)), H.to_host(Vd, (d, d))
                nrm = np.linalg.norm(A, 2)
                wr = np.linalg.eigvalsh(A)[::-1]
                line += (f" | eig {np.abs(w - wr).max() / nrm:.2e} resid {np.linalg.norm(A @ V.T - V.T * w, 2) / nrm:.2e} "
                         f"orth {np.linalg.norm(V @ V.T - np.eye(d), 2):.2e}")
            print(line, flush=True)
        else:
            p, q = (int(t) for t in sz.split("x"))
            r = min(p, q)
            A = rng.standard_normal((p, q))
            Ad, Ud, sd, Vd = H.to_device(A), H.alloc(p * r * 8), H.alloc(r * 8), H.alloc(r * q * 8)
            sw = C.c_int(0)
            ts = []
            for _ in range(reps + 1):
                H.sync()
                t0 = time.perf_counter()
                H.check(H.lib.ccz_gesvj(H.raw, C.c_void_p(Ad.ptr), p, q, C.c_void_p(Ud.ptr), C.c_void_p(sd.ptr), C.c_void_p(Vd.ptr), C.byref(sw)))
                H.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            warm = sorted(ts[1:])
            line = f"gesv {p}x{q}: sweeps {sw.value} first {ts[0]:.2f} ms best {warm[0]:.2f} ms median {warm[len(warm) // 2]:.2f} ms"
            if check:
                s = H.to_host(sd, (r,))
                sr = np.linalg.svd(A, compute_uv=False)
                line += f" | sigma {np.abs(s - sr).max() / sr[0]:.2e}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
