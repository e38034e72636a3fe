"""ctypes binding of ``libccz`` (include/ccz.h) -- the only door to the device.

There is deliberately no CPU fallback: if the HIP library has not been built or
no GPU is visible, every entry point raises.  ``bind`` is also used by the CPU
test-suite to attach the *test double* ``tests/hostsim/libccz_hostsim.so`` (the
product solver drivers compiled against host loops) -- the package itself only
ever loads ``cca_zoo_amd/lib/libccz.so``.
"""

from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np

__all__ = ["CCZError", "Handle", "bind", "library", "library_path", "default_handle", "handle_for", "device_index"]

F32, F64 = 0, 1
_ERRORS = {
    -1: ValueError,
    -2: MemoryError,
    -3: RuntimeError,
    -4: np.linalg.LinAlgError,
    -5: np.linalg.LinAlgError,
    -6: ValueError,
    -7: RuntimeError,
}


class CCZError(RuntimeError):
    """Raised when libccz itself cannot be loaded."""


class View(C.Structure):
    _fields_ = [("data", C.c_void_p), ("cols", C.c_int64), ("ld", C.c_int64)]


class DevInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 128),
        ("arch", C.c_char * 32),
        ("compute_units", C.c_int),
        ("wavefront", C.c_int),
        ("hbm_bytes", C.c_int64),
        ("lds_bytes_per_cu", C.c_int64),
    ]


_vp, _i64, _int, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
_pi64, _pdbl, _pint = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int)

#: every symbol include/ccz.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "ccz_version": (_int, []),
    "ccz_create": (_int, [C.POINTER(_vp), _int]),
    "ccz_destroy": (_int, [_vp]),
    "ccz_last_error": (C.c_char_p, [_vp]),
    "ccz_set_stream": (_int, [_vp, _vp]),
    "ccz_sync": (_int, [_vp]),
    "ccz_stream_acquire": (_int, [_vp, _vp]),
    "ccz_stream_release": (_int, [_vp, _vp]),
    "ccz_stream_adopt": (_int, [_vp, _vp]),
    "ccz_loss_status": (_int, [_vp, _int, _pint, _pint]),
    "ccz_device_info": (_int, [_vp, C.POINTER(DevInfo)]),
    "ccz_dev_alloc": (_int, [_vp, C.POINTER(_vp), C.c_size_t]),
    "ccz_dev_free": (_int, [_vp, _vp]),
    "ccz_memcpy_h2d": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "ccz_memcpy_d2h": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "ccz_memset0": (_int, [_vp, _vp, C.c_size_t]),
    "ccz_moments": (_int, [_vp, _int, C.POINTER(View), _int, _i64, _int, _vp, _int]),
    "ccz_moments_opts": (_int, [_vp, _int, C.POINTER(View), _int, _i64, _int, _vp, _int, _int, _int]),
    "ccz_moments_symmetrize": (_int, [_vp, _vp, _i64]),
    "ccz_moments_pack": (_int, [_vp, _vp, _i64, _vp]),
    "ccz_moments_unpack": (_int, [_vp, _vp, _i64, _vp]),
    "ccz_moments_pack_blocks": (_int, [_vp, _vp, _i64, _pi64, _int, _vp, _int]),
    "ccz_moments_unpack_blocks": (_int, [_vp, _vp, _i64, _pi64, _int, _vp, _int, _vp]),
    "ccz_solve_defer": (_int, [_vp, _vp]),
    "ccz_comm_unique_id": (_int, [_vp, _vp]),
    "ccz_comm_init_rank": (_int, [_vp, _vp, _int, _int]),
    "ccz_comm_init_all": (_int, [C.POINTER(_vp), _int]),
    "ccz_comm_info": (_int, [_vp, _pint, _pint]),
    "ccz_comm_destroy": (_int, [_vp]),
    "ccz_allreduce_sum_f64": (_int, [_vp, _vp, _i64]),
    "ccz_allreduce_sum_f64_multi": (_int, [C.POINTER(_vp), C.POINTER(_vp), _int, _i64]),
    "ccz_moments_last_ms": (_int, [_vp, _pdbl, _pdbl]),
    "ccz_moments_last_pilot": (_int, [_vp, _pint]),
    "ccz_k1_route": (_int, [_vp, _int, _pint]),
    "ccz_pool_trim": (_int, [_vp, C.POINTER(C.c_size_t)]),
    "ccz_loss_last_route": (_int, [_vp, _pint, _pint]),
    "ccz_moments_last_route": (_int, [_vp, _pint, _pdbl, _pdbl, _pdbl]),
    "ccz_rcca_solve": (_int, [_vp, _vp, _i64, _pi64, _pdbl, _int, _int, _vp, _vp, _vp, _pint]),
    "ccz_mcca_solve": (_int, [_vp, _vp, _i64, _pi64, _int, _pdbl, _dbl, _int, _int, _vp, _vp, _vp, _pint]),
    "ccz_gcca_solve": (_int, [_vp, _vp, _i64, _pi64, _int, _pdbl, _pdbl, _dbl, _int, _int, _vp, _vp, _vp, _pint]),
    "ccz_syevj": (_int, [_vp, _vp, _i64, _vp, _vp, _pint]),
    "ccz_gesvj": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _pint]),
    "ccz_gevp_topk": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _vp]),
    "ccz_svd_topk": (_int, [_vp, _vp, _i64, _i64, _int, _vp, _vp, _vp]),
    "ccz_whitener": (_int, [_vp, _vp, _i64, _i64, _dbl, _vp, _vp, _pi64]),
    "ccz_inv_sqrtm": (_int, [_vp, _vp, _i64, _dbl, _vp]),
    "ccz_potrf_lower": (_int, [_vp, _vp, _i64, _i64]),
    "ccz_trsm_right_lower": (_int, [_vp, _int, _i64, _i64, _vp, _i64, _vp, _i64]),
    "ccz_moments_axpby": (_int, [_vp, _i64, _dbl, _vp, _dbl, _vp]),
    "ccz_moments_subset": (_int, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "ccz_gemm_f64": (_int, [_vp, _int, _int, _i64, _i64, _i64, _dbl, _vp, _i64, _vp, _i64, _dbl, _vp, _i64]),
    "ccz_cca_loss": (_int, [_vp, _int, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _dbl, _vp, _vp, _vp, _i64, _i64]),
    "ccz_pair_loss": (_int, [_vp, _int, C.POINTER(View), _int, _i64, _dbl, _vp, C.POINTER(_vp), _pi64]),
    "ccz_moments_exchange": (_int, [_vp, _vp, _i64, _pi64, _int, _i64, _pi64]),
    "ccz_pair_loss_state_bytes": (_i64, [_int, _pi64, _int]),
    "ccz_pair_loss_forward": (_int, [_vp, _int, C.POINTER(View), _int, _i64, _dbl, _vp, _vp]),
    "ccz_pair_loss_backward": (_int, [_vp, _int, C.POINTER(View), _int, _i64, _vp, _vp, C.POINTER(_vp), _pi64]),
    "ccz_cca_loss_moments": (_int, [_vp, _vp, _i64, _i64, _i64, _dbl, C.POINTER(_dbl), _vp, _vp]),
    "ccz_pair_loss_moments": (_int, [_vp, _vp, _i64, _pi64, _int, _dbl, C.POINTER(_dbl), _vp, _vp]),
    "ccz_cholinv": (_int, [_vp, _int, C.POINTER(_vp), _pi64, C.POINTER(_vp), C.POINTER(_vp)]),
    "ccz_randn_fill": (_int, [_vp, _int, _vp, _i64, _i64, _i64, C.c_uint64, _i64, _i64, _dbl, _int]),
    "ccz_gcca_loss_moments": (_int, [_vp, _vp, _i64, _pi64, _int, _dbl, _int, C.POINTER(_dbl), _vp, _vp]),
    "ccz_factor_loadings": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "ccz_transform": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _i64]),
}


def bind(cdll, strict=True):
    """Attach argtypes/restype for every declared symbol; ``strict`` demands all of them."""
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(cdll, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if strict and missing:
        raise CCZError(f"libccz is missing symbols: {missing}")
    return cdll


def library_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libccz.so")


_lib = None
_lock = threading.Lock()


def library():
    """Load (once) the HIP library.  Fails loudly -- there is no fallback path."""
    global _lib
    with _lock:
        if _lib is None:
            # PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64.so.1 with the SAME sonames
            # as /opt/rocm's.  Whichever is mapped first serves the whole process; torch cannot start
            # on top of the system runtime ("No HIP GPUs are available"), the other order works and
            # gives one shared runtime (same null stream, same allocations).  So: torch first.
            # CCZ_TORCHLESS=1 (a ctypes-only deployment of the linear path: NumPy views, ccz_comm_* for sharding) skips
            # the import and runs on /opt/rocm's runtime; importing torch LATER in such a process is not supported.
            if "torch" not in sys.modules and os.environ.get("CCZ_TORCHLESS", "0") != "1":
                try:
                    import torch  # noqa: F401
                except ImportError:  # pragma: no cover - torch-less deployment uses /opt/rocm's runtime
                    pass
            path = library_path()
            if not os.path.exists(path):
                raise CCZError(
                    f"{path} not found: build it with `python -m cca_zoo_amd.csrc.build` "
                    "(hipcc --offload-arch=gfx950).  cca_zoo_amd has no CPU fallback."
                )
            try:
                _lib = bind(C.CDLL(path), strict=True)
            except OSError as e:  # pragma: no cover - depends on the box
                raise CCZError(f"cannot load {path}: {e}") from e
    return _lib


def _ptr(x):
    """void* of a numpy array / int / None."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(x))


class Handle:
    """One device + one stream + scratch pool.  Not thread-safe."""

    def __init__(self, device=0, lib=None):
        self.lib = lib if lib is not None else library()
        h = C.c_void_p()
        rc = self.lib.ccz_create(C.byref(h), int(device))
        if rc != 0:
            raise _ERRORS.get(rc, RuntimeError)(
                f"ccz_create(device={device}) failed with code {rc} "
                "(is an MI355X visible? cca_zoo_amd has no CPU fallback)"
            )
        self._h = h
        self.device = int(device)

    # -- plumbing --------------------------------------------------------------
    def check(self, rc):
        if rc != 0:
            msg = self.lib.ccz_last_error(self._h)
            msg = msg.decode("utf-8", "replace") if msg else ""
            raise _ERRORS.get(rc, RuntimeError)(f"libccz: {msg} (code {rc})")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.ccz_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @property
    def raw(self):
        return self._h

    def set_stream(self, stream_ptr):
        self.check(self.lib.ccz_set_stream(self._h, _ptr(stream_ptr)))

    def sync(self):
        self.check(self.lib.ccz_sync(self._h))

    def acquire(self, stream_ptr):
        """The handle's stream waits (on the device) for ``stream_ptr``'s work so far; the host does not block."""
        self.check(self.lib.ccz_stream_acquire(self._h, C.c_void_p(int(stream_ptr) or None)))

    def release(self, stream_ptr):
        """``stream_ptr`` waits (on the device) for the handle's work so far; the host does not block."""
        self.check(self.lib.ccz_stream_release(self._h, C.c_void_p(int(stream_ptr) or None)))

    def adopt(self, stream_ptr):
        """Enqueue INTO ``stream_ptr`` from now on (until the next ``acquire``): same hardware queue as the caller."""
        self.check(self.lib.ccz_stream_adopt(self._h, C.c_void_p(int(stream_ptr) or None)))

    def loss_status(self, synchronise=False):
        """(view, pivot) of a factorization failure recorded by an earlier ``ccz_cca_loss`` call, or ``None``;
        clears the record."""
        v, p = C.c_int(0), C.c_int(0)
        self.check(self.lib.ccz_loss_status(self._h, 1 if synchronise else 0, C.byref(v), C.byref(p)))
        return (v.value - 1, p.value) if v.value else None

    def device_info(self):
        info = DevInfo()
        self.check(self.lib.ccz_device_info(self._h, C.byref(info)))
        return {
            "name": info.name.decode(), "arch": info.arch.decode(),
            "compute_units": info.compute_units, "wavefront": info.wavefront,
            "hbm_bytes": info.hbm_bytes, "lds_bytes_per_cu": info.lds_bytes_per_cu,
        }

    # -- device memory ------------------------------------------------------------
    def alloc(self, nbytes):
        p = C.c_void_p()
        self.check(self.lib.ccz_dev_alloc(self._h, C.byref(p), int(nbytes)))
        return DeviceBuffer(self, p.value, int(nbytes))

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        buf = self.alloc(max(arr.nbytes, 8))
        self.check(self.lib.ccz_memcpy_h2d(self._h, _ptr(buf.ptr), _ptr(arr), arr.nbytes))
        buf.shape, buf.dtype = arr.shape, arr.dtype
        return buf

    def h2d(self, dst_ptr, arr):
        """Copy a host array into device memory at ``dst_ptr`` (any device address, e.g. inside a moments buffer)."""
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.ccz_memcpy_h2d(self._h, _ptr(dst_ptr), _ptr(arr), arr.nbytes))

    def memset0(self, ptr, nbytes):
        self.check(self.lib.ccz_memset0(self._h, _ptr(ptr), int(nbytes)))

    def to_host(self, buf, shape, dtype=np.float64, offset_bytes=0):
        out = np.empty(shape, dtype=dtype)
        self.check(self.lib.ccz_memcpy_d2h(self._h, _ptr(out), _ptr(buf.ptr + offset_bytes if isinstance(buf, DeviceBuffer) else int(buf) + offset_bytes), out.nbytes))
        return out

    # -- K1 ---------------------------------------------------------------------------
    def moments(self, views, n_rows, dtype, on_device, moments_ptr, accumulate=False, pilot=None, timed=True):
        """views: list of (ptr_or_ndarray, cols, ld).  ``pilot`` (None: automatic, False: never, True: always shift
        fp32 views) and ``timed=False`` select ``ccz_moments_opts`` -- with ``pilot=True, timed=False`` the call only
        enqueues work."""
        arr = (View * len(views))()
        keep = []
        for i, (data, cols, ld) in enumerate(views):
            if isinstance(data, np.ndarray):
                keep.append(data)
                arr[i].data = data.ctypes.data
            else:
                arr[i].data = int(data)
            arr[i].cols, arr[i].ld = int(cols), int(ld)
        if pilot is None and timed:
            self.check(self.lib.ccz_moments(self._h, int(dtype), arr, len(views), int(n_rows),
                                            1 if on_device else 0, _ptr(moments_ptr), 1 if accumulate else 0))
        else:
            mode = 1 if pilot is None else (2 if pilot else 0)
            self.check(self.lib.ccz_moments_opts(self._h, int(dtype), arr, len(views), int(n_rows), 1 if on_device else 0,
                                                 _ptr(moments_ptr), 1 if accumulate else 0, mode, 1 if timed else 0))

    def moments_symmetrize(self, moments_ptr, D):
        self.check(self.lib.ccz_moments_symmetrize(self._h, _ptr(moments_ptr), int(D)))

    def moments_pack(self, moments_ptr, D, packed_ptr):
        self.check(self.lib.ccz_moments_pack(self._h, _ptr(moments_ptr), int(D), _ptr(packed_ptr)))

    def moments_unpack(self, packed_ptr, D, moments_ptr):
        self.check(self.lib.ccz_moments_unpack(self._h, _ptr(packed_ptr), int(D), _ptr(moments_ptr)))

    HEAD, TAIL, BOTH = 1, 2, 3

    def moments_pack_blocks(self, moments_ptr, D, dims, packed_ptr, which=3):
        """moments -> blocks layout ``[diag-block triangles | colsum | n slot || off-diagonal blocks]`` (ccz.h)."""
        da = (C.c_int64 * len(dims))(*[int(d) for d in dims])
        self.check(self.lib.ccz_moments_pack_blocks(self._h, _ptr(moments_ptr), int(D), da, len(dims), _ptr(packed_ptr), int(which)))

    def moments_unpack_blocks(self, packed_ptr, D, dims, moments_ptr, which=3, on_stream=None):
        da = (C.c_int64 * len(dims))(*[int(d) for d in dims])
        self.check(self.lib.ccz_moments_unpack_blocks(self._h, _ptr(packed_ptr), int(D), da, len(dims), _ptr(moments_ptr), int(which),
                                                      C.c_void_p(int(on_stream)) if on_stream else None))

    # -- the exchange step behind the ABI (RCCL; no torch.distributed needed) -------------------
    def comm_unique_id(self) -> bytes:
        """128 opaque bytes (rank 0 creates them; every rank passes the same bytes to ``comm_init_rank``)."""
        buf = C.create_string_buffer(128)
        self.check(self.lib.ccz_comm_unique_id(self._h, C.cast(buf, C.c_void_p)))
        return buf.raw

    def comm_init_rank(self, unique_id: bytes, world: int, rank: int):
        if len(unique_id) != 128:
            raise ValueError("the communicator id is 128 bytes (Handle.comm_unique_id())")
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self.check(self.lib.ccz_comm_init_rank(self._h, C.cast(buf, C.c_void_p), int(world), int(rank)))

    def comm_info(self):
        """(world size, rank) of the handle's communicator; (0, -1) without one."""
        w, r = C.c_int(0), C.c_int(-1)
        self.check(self.lib.ccz_comm_info(self._h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def comm_destroy(self):
        self.check(self.lib.ccz_comm_destroy(self._h))

    def allreduce_sum_f64(self, ptr, count):
        """In-place float64 SUM over the communicator's ranks, enqueued on the handle's stream."""
        self.check(self.lib.ccz_allreduce_sum_f64(self._h, _ptr(ptr), int(count)))

    def moments_exchange(self, moments_ptr, D, dims, n_local):
        """The whole exchange step behind the ABI (``ccz_moments_exchange``): returns the global row count; the off-diagonal
        blocks may still be in flight -- the next solve waits for them on the device."""
        n_total = C.c_int64(0)
        dims_a = (C.c_int64 * len(dims))(*[int(d) for d in dims])
        self.check(self.lib.ccz_moments_exchange(self._h, C.c_void_p(int(moments_ptr)), int(D), dims_a, len(dims), int(n_local),
                                                 C.byref(n_total)))
        return int(n_total.value)

    def solve_defer(self, event_ptr):
        """The next ``*_solve`` waits (on the device) for this hipEvent before reading off-diagonal moment blocks."""
        self.check(self.lib.ccz_solve_defer(self._h, C.c_void_p(int(event_ptr)) if event_ptr else None))

    def moments_axpby(self, D, alpha, x_ptr, beta, y_ptr):
        """y <- alpha x + beta y over two moment buffers (moments are additive over disjoint row sets)."""
        self.check(self.lib.ccz_moments_axpby(self._h, int(D), float(alpha), _ptr(x_ptr), float(beta), _ptr(y_ptr)))

    def moments_subset(self, moments_ptr, D, col0, D_sub, out_ptr):
        self.check(self.lib.ccz_moments_subset(self._h, _ptr(moments_ptr), int(D), int(col0), int(D_sub), _ptr(out_ptr)))

    def gemm(self, tA, tB, M, N, K, alpha, A_ptr, lda, B_ptr, ldb, beta, C_ptr, ldc):
        self.check(self.lib.ccz_gemm_f64(self._h, int(tA), int(tB), int(M), int(N), int(K), float(alpha), _ptr(A_ptr), int(lda),
                                         _ptr(B_ptr), int(ldb), float(beta), _ptr(C_ptr), int(ldc)))

    def moments_last_ms(self):
        g, s = C.c_double(), C.c_double()
        self.check(self.lib.ccz_moments_last_ms(self._h, C.byref(g), C.byref(s)))
        return g.value, s.value

    def moments_last_pilot(self):
        """True if the last K1 launch on this handle used the pilot-mean (shifted) fp32 Gram kernel."""
        u = C.c_int(0)
        self.check(self.lib.ccz_moments_last_pilot(self._h, C.byref(u)))
        return bool(u.value)

    def pool_trim(self):
        """Return the handle's cached scratch blocks to the driver (bytes released)."""
        n = C.c_size_t(0)
        self.check(self.lib.ccz_pool_trim(self._h, C.byref(n)))
        return int(n.value)

    K1_ROUTES = {"auto": 0, "fp32": 1, "bf16x2": 2, "fp64": 3}

    def k1_route(self, route=None):
        """Arithmetic route of fp32 views through K1 (include/ccz.h: CCZ_K1_*): "auto" | "fp32" | "bf16x2".  Returns the
        previous setting's name; ``None`` only queries."""
        prev = C.c_int(0)
        code = -1 if route is None else (self.K1_ROUTES[route] if isinstance(route, str) else int(route))
        self.check(self.lib.ccz_k1_route(self._h, code, C.byref(prev)))
        return {v: k for k, v in self.K1_ROUTES.items()}[prev.value]

    def moments_last_route(self):
        """(route name, split_ms, mfma_ms, reduce_ms) of the last K1 launch on this handle (stage times: timed bf16x2 launches)."""
        r, a, b, d = C.c_int(0), C.c_double(), C.c_double(), C.c_double()
        self.check(self.lib.ccz_moments_last_route(self._h, C.byref(r), C.byref(a), C.byref(b), C.byref(d)))
        return {v: k for k, v in self.K1_ROUTES.items()}.get(r.value, "none"), a.value, b.value, d.value

    def loss_last_route(self):
        """(forward K1 route, backward product route) of the last loss on this handle, by name ("none" before the first)."""
        f, b = C.c_int(0), C.c_int(0)
        self.check(self.lib.ccz_loss_last_route(self._h, C.byref(f), C.byref(b)))
        names = {v: k for k, v in self.K1_ROUTES.items()}
        names[0] = "none"
        return names.get(f.value, "none"), names.get(b.value, "none")

    # -- fused solves ----------------------------------------------------------------------
    def _solve_out(self, dims, k):
        D = int(sum(dims))
        kk = int(min(k, D))
        return np.zeros(D * kk), np.zeros(D), np.zeros(kk), C.c_int(0)

    @staticmethod
    def _split(W, dims, kk):
        out, o = [], 0
        for d in dims:
            out.append(W[o:o + d * kk].reshape(d, kk).copy())
            o += d * kk
        return out

    def rcca_solve(self, moments_ptr, n, dims, c, center, k):
        dims_a = (C.c_int64 * 2)(*[int(d) for d in dims])
        c_a = (C.c_double * 2)(*[float(v) for v in c])
        W, mu, vals, kout = self._solve_out(dims, k)
        self.check(self.lib.ccz_rcca_solve(self._h, _ptr(moments_ptr), int(n), dims_a, c_a, int(bool(center)),
                                           int(k), _ptr(W), _ptr(mu), _ptr(vals), C.byref(kout)))
        kk = kout.value
        return self._split(W, dims, kk), np.split(mu, np.cumsum(dims)[:-1]), vals[:kk]

    def mcca_solve(self, moments_ptr, n, dims, c, eps, center, k):
        m = len(dims)
        dims_a = (C.c_int64 * m)(*[int(d) for d in dims])
        c_a = (C.c_double * m)(*[float(v) for v in c])
        W, mu, vals, kout = self._solve_out(dims, k)
        self.check(self.lib.ccz_mcca_solve(self._h, _ptr(moments_ptr), int(n), dims_a, m, c_a, float(eps),
                                           int(bool(center)), int(k), _ptr(W), _ptr(mu), _ptr(vals), C.byref(kout)))
        kk = kout.value
        return self._split(W, dims, kk), np.split(mu, np.cumsum(dims)[:-1]), vals[:kk]

    def gcca_solve(self, moments_ptr, n, dims, c, view_weights, eps, center, k):
        m = len(dims)
        dims_a = (C.c_int64 * m)(*[int(d) for d in dims])
        c_a = (C.c_double * m)(*[float(v) for v in c])
        w_a = (C.c_double * m)(*[float(v) for v in view_weights])
        W, mu, vals, kout = self._solve_out(dims, k)
        self.check(self.lib.ccz_gcca_solve(self._h, _ptr(moments_ptr), int(n), dims_a, m, c_a, w_a, float(eps),
                                           int(bool(center)), int(k), _ptr(W), _ptr(mu), _ptr(vals), C.byref(kout)))
        kk = kout.value
        return self._split(W, dims, kk), np.split(mu, np.cumsum(dims)[:-1]), vals[:kk]


class DeviceBuffer:
    """Owning wrapper of a ``ccz_dev_alloc`` allocation."""

    def __init__(self, handle, ptr, nbytes):
        self.handle, self.ptr, self.nbytes = handle, ptr, nbytes
        self.shape, self.dtype = None, None

    def free(self):
        if self.ptr and self.handle is not None and getattr(self.handle, "_h", None):
            self.handle.lib.ccz_dev_free(self.handle._h, C.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass

    def __int__(self):
        return int(self.ptr)


_default = {}


def default_handle(device=None):
    """Process-wide handle per device (created on first use)."""
    if device is None:
        device = int(os.environ.get("CCZ_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    h = _default.get(device)
    if h is None:
        h = _default[device] = Handle(device)
    return h


def device_index(tensor) -> int:
    """Ordinal of the GPU a CUDA tensor lives on (``cuda`` without an index = torch's current device)."""
    idx = tensor.device.index
    if idx is None:
        import torch

        idx = torch.cuda.current_device()
    return int(idx)


def handle_for(arrays):
    """The handle every libccz call on ``arrays`` must go through: bound to the GPU the CUDA tensors among them
    live on (all on ONE device, else ``ValueError``), the process default for host arrays.  A handle launches
    on its own device only -- a device-0 handle reading ``cuda:1`` pointers is a fault (or silent peer access)."""
    devs = set()
    for a in arrays:
        if type(a).__module__.startswith("torch") and getattr(a, "is_cuda", False):
            devs.add(device_index(a))
    if len(devs) > 1:
        raise ValueError(f"all CUDA views must live on the same device, got devices {sorted(devs)}")
    if devs:
        return default_handle(devs.pop())
    return default_handle()
