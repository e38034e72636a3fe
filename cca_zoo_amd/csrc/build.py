#!/usr/bin/env python3
"""Build ``cca_zoo_amd/lib/libccz.so`` for gfx950 with hipcc (cross-compiles without a GPU).

    python -m cca_zoo_amd.csrc.build [--force] [--verbose]

Sources are compiled one object per file (incremental by mtime) and linked into a
single shared library whose exported symbols are exactly those of ``include/ccz.h``.
"""

from __future__ import annotations

import argparse
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OBJ = os.path.join(HERE, "_obj")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libccz.so")
ARCH = "gfx950"
SOURCES = ["api.hip", "comm.hip", "ops_hip.hip", "gram.hip", "gram_split.hip", "gemm_split.hip", "project_split.hip", "gemm_big.hip", "gemm64_big.hip", "gemm64_skinny.hip", "cholinv.hip", "evd_block.hip", "loss.hip", "rng.hip", "solve.cpp"]
HEADERS = ["ops.h", "hip_common.h", "gram_map.h", "split_mma.h", "jacobi_dev.h", "rng_hash.h", os.path.join(ROOT, "include", "ccz.h")]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
HIPFLAGS = [f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-fgpu-rdc" if False else "-fno-gpu-rdc"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (ROCm toolchain required to build libccz)")
    return exe


def _newest_header():
    t = 0.0
    for h in HEADERS:
        p = h if os.path.isabs(h) else os.path.join(HERE, h)
        t = max(t, os.path.getmtime(p))
    return max(t, os.path.getmtime(os.path.abspath(__file__)))


LAST_ACTION = ""   # what the last build() did: "rebuilt N of M sources (...)" / "up to date (...)"


def build(force=False, verbose=False):
    """Bring ``lib/libccz.so`` up to date.  ``CCZ_FORCE_BUILD=1`` (or ``force``) recompiles everything; otherwise the
    library is rebuilt when it is missing or OLDER than any source or header -- a shipped binary that is newer than the
    whole source tree is used as it is (and said so), even where the per-file objects did not travel with it."""
    global LAST_ACTION
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    force = force or os.environ.get("CCZ_FORCE_BUILD", "") not in ("", "0")
    hdr_t = _newest_header()
    src_t = max(os.path.getmtime(os.path.join(HERE, s)) for s in SOURCES)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(hdr_t, src_t):
        LAST_ACTION = f"up to date: {os.path.relpath(LIB, ROOT)} is newer than every source and header (CCZ_FORCE_BUILD=1 rebuilds)"
        if verbose:
            print(LAST_ACTION, flush=True)
        return LIB
    cc = hipcc()
    objs, rebuilt, compiled = [], False, []
    for src in SOURCES:
        sp = os.path.join(HERE, src)
        op = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        if not force and os.path.exists(op) and os.path.getmtime(op) >= max(os.path.getmtime(sp), hdr_t):
            continue
        cmd = [cc] + COMMON
        if src.endswith(".hip"):
            cmd += HIPFLAGS
        else:
            cmd += ["-x", "c++"]
        cmd += ["-c", sp, "-o", op]
        t0 = time.time()
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        rebuilt = True
        compiled.append(src)
        if verbose:
            print(f"  {src}: {time.time() - t0:.1f}s", flush=True)
    if rebuilt or force or not os.path.exists(LIB):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    LAST_ACTION = f"rebuilt {len(compiled)} of {len(SOURCES)} sources ({', '.join(compiled) or 'link only'}){' [forced]' if force else ''}"
    return LIB


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", "-v", action="store_true")
    a = ap.parse_args(argv)
    print(build(a.force, a.verbose))


if __name__ == "__main__":
    main()
