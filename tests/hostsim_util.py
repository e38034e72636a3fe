"""Build + load the host test double of libccz's solver drivers (CPU tests only)."""

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SIM_DIR = os.path.join(HERE, "hostsim")
SIM_SO = os.path.join(SIM_DIR, "libccz_hostsim.so")
SOURCES = [os.path.join(SIM_DIR, "ops_host.cpp"), os.path.join(ROOT, "cca_zoo_amd", "csrc", "solve.cpp")]
HEADERS = [os.path.join(ROOT, "cca_zoo_amd", "csrc", "ops.h"), os.path.join(ROOT, "cca_zoo_amd", "csrc", "rng_hash.h"),
           os.path.join(ROOT, "include", "ccz.h")]


SIM_SO_SAN = os.path.join(SIM_DIR, "libccz_hostsim_asan.so")


def sanitized():
    """CCZ_HOSTSIM_SANITIZE=1: the double is built with AddressSanitizer + UndefinedBehaviorSanitizer (the
    product driver source solve.cpp runs under both).  The process must have libasan preloaded: tests/test_sanitizers.py."""
    return os.environ.get("CCZ_HOSTSIM_SANITIZE", "") not in ("", "0")


def build_hostsim():
    newest = max(os.path.getmtime(p) for p in SOURCES + HEADERS)
    so = SIM_SO_SAN if sanitized() else SIM_SO
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        flags = (["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
                 if sanitized() else ["-O2"])
        subprocess.check_call(["g++"] + flags + ["-std=c++17", "-fPIC", "-shared", "-o", so] + SOURCES)
    return so


def hostsim_handle():
    from cca_zoo_amd import _backend

    import torch

    lib = _backend.bind(ctypes.CDLL(build_hostsim()), strict=False)
    h = _backend.Handle(0, lib=lib)
    h.torch_device = torch.device("cpu")          # the double's "device" memory is host memory
    return h


def pack_moments(G, s):
    return np.concatenate([np.ascontiguousarray(G, dtype=np.float64).ravel(), np.asarray(s, dtype=np.float64)])
