#!/usr/bin/env python3
"""Why does every other solve of the bench loop run ~10 ms slower on a box that has been under load?

One process = one setting of the read-once switches (CCZ_POTRF_LOOKAHEAD, CCZ_GRAPHS, ...).  Inside it the SAME data and
model go through a few loop shapes, each printing one line per fit (K1 by HIP events, solve wall, shader / memory clock
and package power read from sysfs right after K1 and right after the solve):

    fit      K1 (n rows) -> solve, back to back                       (bench.py's loop)
    short    K1 on the first `--short-rows` rows -> solve              (changes the loop period, not the parity)
    gap      K1 -> host sleep `--gap-ms` -> solve                      (idle time between the MFMA burst and the chain)
    solve    solve only, moments kept                                  (no MFMA burst at all)
    two      K1, K1 -> solve                                           (two bursts per solve)

    python tools/period2_probe.py --preheat-s 60 --fits 14 --modes fit,short,gap,solve,two
"""
import argparse
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sysfs_state():
    out = {}
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")[:1]:
        try:
            out["sclk"] = [l.split(":")[1].replace("*", "").strip() for l in open(p) if "*" in l][0]
        except Exception:
            pass
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_mclk")[:1]:
        try:
            out["mclk"] = [l.split(":")[1].replace("*", "").strip() for l in open(p) if "*" in l][0]
        except Exception:
            pass
    for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")[:1] or \
            glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")[:1]:
        try:
            out["W"] = int(open(p).read()) // 1000000
        except Exception:
            pass
    for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/temp1_input")[:1]:
        try:
            out["C"] = int(open(p).read()) // 1000
        except Exception:
            pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--short-rows", type=int, default=620_000)
    ap.add_argument("--gap-ms", type=float, default=60.0)
    ap.add_argument("--fits", type=int, default=14)
    ap.add_argument("--preheat-s", type=float, default=0.0)
    ap.add_argument("--modes", default="fit,short,gap,solve,two")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    import numpy as np
    import torch

    from cca_zoo_amd import _backend
    from cca_zoo_amd._moments import compute_moments
    from cca_zoo_amd.datasets import JointData

    d, k = 4096, 64
    h = _backend.default_handle(0)
    jd = JointData(n_views=2, n_samples=a.rows, latent_dimensions=k, n_features=[d, d], random_state=0,
                   latent_scales=list(np.linspace(2.0, 0.5, k)))
    views = jd.sample_device(device="cuda:0", dtype=torch.float32, n_samples=a.rows, seed=20260)
    torch.cuda.synchronize()
    tag = a.tag or ",".join(f"{k_}={v}" for k_, v in os.environ.items() if k_.startswith("CCZ_")) or "default"
    print(f"## {tag}: sysfs at start {sysfs_state()}", flush=True)

    def k1(vs):
        mom, keep, nt, dims, kind = compute_moments(vs, h)
        h.sync()
        return mom, keep, nt, dims

    def solve(mom, nt, dims):
        t0 = time.perf_counter()
        h.rcca_solve(mom, nt, dims, [0.0, 0.0], True, k)
        return (time.perf_counter() - t0) * 1e3

    t_end = time.perf_counter() + a.preheat_s
    nheat = 0
    while time.perf_counter() < t_end:
        mom, keep, nt, dims = k1(views)
        solve(mom, nt, dims)
        del keep
        nheat += 1
    if nheat:
        print(f"## preheated with {nheat} fits; sysfs {sysfs_state()}", flush=True)

    short = [v[:a.short_rows] for v in views]
    for mode in a.modes.split(","):
        series = []
        mom, keep, nt, dims = k1(views)
        for it in range(a.fits):
            t0 = time.perf_counter()
            if mode in ("fit", "gap", "two"):
                del keep
                mom, keep, nt, dims = k1(views)
                if mode == "two":
                    del keep
                    mom, keep, nt, dims = k1(views)
            elif mode == "short":
                del keep
                mom, keep, nt, dims = k1(short)
            g_ms = h.moments_last_ms()[0]
            s1 = sysfs_state()
            if mode == "gap":
                time.sleep(a.gap_ms * 1e-3)
            sv = solve(mom, nt, dims)
            s2 = sysfs_state()
            series.append(sv)
            print(f"{tag} {mode:5s} it {it:2d}: step {(time.perf_counter() - t0) * 1e3:7.1f}  gram {g_ms:6.1f}  solve {sv:6.1f}  "
                  f"afterK1 {s1}  afterSolve {s2}", flush=True)
        s = np.array(series)
        even, odd = s[0::2], s[1::2]
        print(f"## {tag} {mode}: solve min {s.min():.1f} mean {s.mean():.1f} max {s.max():.1f} | even-index mean {even.mean():.1f} "
              f"odd-index mean {odd.mean():.1f}", flush=True)
        del keep


if __name__ == "__main__":
    main()
