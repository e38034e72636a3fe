"""DCCA correlation losses as ``nn.Module`` callables backed by libccz.

Reference: cca_zoo/deep/objectives.py:9-21 (``_inv_sqrtm``), :24-102 (``CCALoss``),
:105-153 (``MCCALoss``).  Contract kept: ``forward(list[Tensor (batch x d_i)]) ->
0-dim Tensor`` on the inputs' device/dtype, differentiable w.r.t. every input,
``ValueError`` matching "exactly 2" for a wrong number of views, stateless modules.

What runs underneath (``ccz_cca_loss``): one fused MFMA pass for the batch second
moments, float64 Cholesky solves on the d x d blocks, ``loss = -tr(S11^-1 S12 S22^-1
S21)`` -- identical to the reference's ``-sum eigvalsh(T'T)`` because the ``+eps I``
makes the eigenvalue clamp inactive -- and the closed-form
input gradients (row 9) as four MFMA GEMMs, instead of autograd through ``eigh``
(which is NaN at repeated eigenvalues; the closed form is not).

Inputs must be CUDA tensors: there is no CPU implementation in this package.

Stream contract (the reference's ``training_step``: cca_zoo/deep/_base.py:78-104).  The loss is an
ordinary node of the autograd graph: libccz's stream is joined to ``torch.cuda.current_stream()`` on the DEVICE before and
after the call (``ccz_stream_acquire`` / ``ccz_stream_release``; nothing at all for the default stream, with which the
handle's blocking stream is ordered implicitly).  ``CCALoss`` / ``MCCALoss`` never make the host wait: a failed
factorization (possible only with eps <= 0 or non-finite inputs) turns the loss into NaN and is raised as ``LinAlgError``
by the NEXT loss call on that device or by :func:`check_async_errors`.
"""

from __future__ import annotations

import ctypes as C

import torch
from torch.autograd.function import once_differentiable
import torch.nn as nn

from cca_zoo_amd import _backend


def _stream_ptr(t: torch.Tensor) -> int:
    return int(torch.cuda.current_stream(t.device).cuda_stream)


def _raise_pending(h, synchronise: bool = False) -> None:
    st = h.loss_status(synchronise)
    if st is not None:
        import numpy as np

        raise np.linalg.LinAlgError(
            f"an earlier CCALoss / MCCALoss call on this device returned NaN: S_{st[0] + 1}{st[0] + 1} + eps I was not "
            f"positive definite (pivot {st[1]}); eps must be > 0 and the representations finite")


def check_async_errors(device=None) -> None:
    """Drain libccz's stream on ``device`` (default: the current one) and raise ``LinAlgError`` if a loss evaluated since the
    last check met a non-positive pivot (its value was NaN).  The losses themselves never wait for the device."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    _raise_pending(_backend.default_handle(dev if dev is not None else torch.cuda.current_device()), synchronise=True)


def _view_of(t: torch.Tensor) -> torch.Tensor:
    """Row-major with unit column stride and non-overlapping rows (else a contiguous copy)."""
    return t if (t.stride(1) == 1 and t.stride(0) >= t.shape[1]) else t.contiguous()


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: expected a CUDA (ROCm) tensor; cca_zoo_amd runs on the MI355X only "
            "and has no CPU fallback"
        )
    if t.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"{what}: dtype must be float32 or float64, got {t.dtype}")


class _ShardedCCALossFn(torch.autograd.Function):
    """CCALoss of a batch whose rows are spread over the ranks of ``row_sharded()``: K1 on the local rows, ONE
    all-reduce of the packed (d1+d2)^2 moments, the d x d solve replicated on every rank
    (``ccz_cca_loss_moments``) and the gradient of the GLOBAL loss with respect to the LOCAL rows as one
    ``ccz_transform`` GEMM -- no all-gather of the embeddings."""

    @staticmethod
    def forward(ctx, z1: torch.Tensor, z2: torch.Tensor, eps: float) -> torch.Tensor:
        from cca_zoo_amd import _dist

        _require_cuda(z1, "CCALoss")
        _require_cuda(z2, "CCALoss")
        if z1.dim() != 2 or z2.dim() != 2 or z1.shape[0] != z2.shape[0]:
            raise ValueError("CCALoss expects two (batch, d_i) tensors with equal batch size")
        dt = z1.dtype
        zcat = torch.cat([z1, z2.to(dt)], dim=1).contiguous()
        n_local, D = int(zcat.shape[0]), int(zcat.shape[1])
        d1, d2 = int(z1.shape[1]), int(z2.shape[1])
        dev = zcat.device
        h = _backend.handle_for([zcat])
        mom = torch.empty(D * D + D, dtype=torch.float64, device=dev)
        sp = _stream_ptr(zcat)
        h.acquire(sp)
        h.moments([(zcat.data_ptr(), D, D)], n_local, _backend.F32 if dt == torch.float32 else _backend.F64, True, mom.data_ptr(),
                  pilot=True, timed=False)
        npk = D * (D + 1) // 2 + D
        packed = torch.empty(npk + 1, dtype=torch.float64, device=dev)
        h.moments_pack(mom.data_ptr(), D, packed.data_ptr())
        h.release(sp)                                      # the collective (torch's stream) follows libccz's stream on the device
        packed[npk:].fill_(float(n_local))
        n_total = _dist.allreduce_moments(packed, _dist.active_group())
        h.acquire(sp)
        h.moments_unpack(packed.data_ptr(), D, mom.data_ptr())
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        loss_h = C.c_double(0.0)
        gamma = torch.empty((D, D), dtype=torch.float64, device=dev) if need else None
        mean = torch.empty(D, dtype=torch.float64, device=dev) if need else None
        h.check(h.lib.ccz_cca_loss_moments(h.raw, C.c_void_p(mom.data_ptr()), int(n_total), d1, d2, float(eps),
                                           C.byref(loss_h), C.c_void_p(gamma.data_ptr()) if need else None,
                                           C.c_void_p(mean.data_ptr()) if need else None))
        if need:
            g = _project(zcat, mean, gamma)
            ctx.save_for_backward(g)
            ctx.split = (d1, d2)
            ctx.dtypes = (z1.dtype, z2.dtype)
        return torch.tensor(loss_h.value, dtype=dt, device=dev)

    @staticmethod
    def backward(ctx, grad_out):
        (g,) = ctx.saved_tensors
        g1, g2 = torch.split(g, ctx.split, dim=1)
        return (grad_out * g1).to(ctx.dtypes[0]), (grad_out * g2).to(ctx.dtypes[1]), None


class _PairLossFn(torch.autograd.Function):
    """Sum over all view pairs of the CCA loss of ONE batch (two views: ``CCALoss``) as a two-phase autograd node:
    ``ccz_pair_loss_forward`` -- one K1 pass over ``[z_1 .. z_m]``, one Cholesky + inverse per VIEW (the reference re-centres
    every view and recomputes its ``S_aa^-1/2`` once per PAIR, cca_zoo/deep/objectives.py:138-153), the loss written on the
    device, ``Gamma`` left in a small state tensor -- and ``ccz_pair_loss_backward``, which forms every view's gradient
    ``grad_out * (Z - mean) Gamma`` from the views where they lie, the upstream gradient applied inside the product (no
    ``grad_out * g`` passes over the n x d gradients afterwards).  Enqueue-only: no host synchronisation, no concatenated
    copy."""

    @staticmethod
    def forward(ctx, eps: float, what: str, *zs: torch.Tensor) -> torch.Tensor:
        for z in zs:
            _require_cuda(z, what)
            if z.dim() != 2 or z.shape[0] != zs[0].shape[0]:
                raise ValueError(f"{what} expects (batch, d_i) tensors with equal batch size")
        dt = zs[0].dtype
        vs = [_view_of(z if z.dtype == dt else z.to(dt)) for z in zs]
        m = len(vs)
        h = _backend.handle_for(vs)
        _raise_pending(h)
        need = any(ctx.needs_input_grad[2:])
        loss = torch.empty((), dtype=dt, device=vs[0].device)
        code = _backend.F32 if dt == torch.float32 else _backend.F64
        state = None
        if need:
            dims = (C.c_int64 * m)(*[int(v.shape[1]) for v in vs])
            nbytes = int(h.lib.ccz_pair_loss_state_bytes(code, dims, m))
            if nbytes <= 0:
                raise ValueError(f"{what}: unsupported views")
            state = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=vs[0].device)
        views = _views_of(vs)
        sp = _stream_ptr(vs[0])
        h.adopt(sp)                                         # the loss's kernels go INTO torch's current stream
        try:
            h.check(h.lib.ccz_pair_loss_forward(h.raw, code, views, m, int(vs[0].shape[0]), float(eps), C.c_void_p(loss.data_ptr()),
                                                C.c_void_p(state.data_ptr()) if need else None))
        finally:
            # whatever happened (ENOTSPD on the wide route, EINVAL, out of memory): the handle goes HOME to its own stream,
            # ordered behind the caller's -- a side stream may be destroyed later, and entry points that do not acquire
            # (host-array fits, gevp, sync) must never run on the legacy null stream by accident
            h.acquire(sp)
        if need:
            ctx.save_for_backward(state, *vs)
            ctx.dtypes = [z.dtype for z in zs]
            ctx.wanted = list(ctx.needs_input_grad[2:])
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        state, *vs = ctx.saved_tensors
        m = len(vs)
        dt = vs[0].dtype
        h = _backend.handle_for(vs)
        go = grad_out.detach().to(device=vs[0].device, dtype=dt).reshape(()).contiguous()
        grads = [torch.empty_like(v, memory_format=torch.contiguous_format) if w else None for v, w in zip(vs, ctx.wanted)]
        gp = (C.c_void_p * m)(*[g.data_ptr() if g is not None else None for g in grads])
        ldg = (C.c_int64 * m)(*[int(g.stride(0)) if g is not None else 0 for g in grads])
        views = _views_of(vs)
        sp = _stream_ptr(vs[0])
        h.adopt(sp)
        try:
            h.check(h.lib.ccz_pair_loss_backward(h.raw, _backend.F32 if dt == torch.float32 else _backend.F64, views, m, int(vs[0].shape[0]),
                                                 C.c_void_p(state.data_ptr()), C.c_void_p(go.data_ptr()), gp, ldg))
        finally:
            h.acquire(sp)
        return (None, None, *[g if g is None else g.to(t) for g, t in zip(grads, ctx.dtypes)])


def _views_of(vs):
    views = (_backend.View * len(vs))()
    for i, v in enumerate(vs):
        views[i].data, views[i].cols, views[i].ld = v.data_ptr(), int(v.shape[1]), int(v.stride(0))
    return views


class _InvSqrtmFn(torch.autograd.Function):
    """``A^-1/2`` with eigenvalues clamped at ``eps`` as a differentiable node (the reference's ``_inv_sqrtm`` sits inside
    the autograd graph of ``CCALoss.forward``, cca_zoo/deep/objectives.py:9-21, :94-97): ``A = V diag(lam) V'`` by the device
    Jacobi EVD (``ccz_syevj``), ``F = V diag(f(lam)) V'`` with ``f = max(., eps)^-1/2``, and the Daleckii-Krein backward
    ``dA = V ((V' G V) o K) V'``, ``K_ij = (f_i - f_j) / (lam_i - lam_j)`` (``f'(lam_i)`` on the diagonal and between equal
    eigenvalues -- finite where ``eigh``'s own backward divides by zero).  All products are ``ccz_gemm_f64``."""

    @staticmethod
    def forward(ctx, A: torch.Tensor, eps: float) -> torch.Tensor:
        h = _backend.handle_for([A])
        d = int(A.shape[0])
        a64 = A.detach().to(torch.float64).contiguous().clone()      # ccz_syevj overwrites its input
        w = torch.empty(d, dtype=torch.float64, device=A.device)
        Vr = torch.empty((d, d), dtype=torch.float64, device=A.device)   # row i = eigenvector i
        sp = _stream_ptr(A)
        h.acquire(sp)
        try:                                                         # (ENOCONV / out of memory must not leave torch's stream un-joined)
            h.check(h.lib.ccz_syevj(h.raw, C.c_void_p(a64.data_ptr()), d, C.c_void_p(w.data_ptr()), C.c_void_p(Vr.data_ptr()), None))
        finally:
            h.release(sp)
        f = torch.clamp(w, min=eps).rsqrt()
        scaled = (Vr * f[:, None]).contiguous()                      # diag(f) V'
        out = torch.empty((d, d), dtype=torch.float64, device=A.device)
        h.acquire(sp)
        try:
            h.gemm(True, False, d, d, d, 1.0, Vr.data_ptr(), d, scaled.data_ptr(), d, 0.0, out.data_ptr(), d)   # V diag(f) V'
        finally:
            h.release(sp)
        ctx.save_for_backward(w, Vr, f)
        ctx.eps, ctx.dtype = float(eps), A.dtype
        return out.to(A.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, G):
        w, Vr, f = ctx.saved_tensors
        d = int(w.shape[0])
        h = _backend.handle_for([Vr])
        g64 = G.detach().to(torch.float64)
        g64 = (0.5 * (g64 + g64.t())).contiguous()                   # F is symmetric: only the symmetric part of G acts
        fp = torch.where(w > ctx.eps, -0.5 * f ** 3, torch.zeros_like(f))     # f'(lam); 0 where the clamp is active
        dl = w[:, None] - w[None, :]
        close = dl.abs() <= 1e-12 * torch.clamp(w.abs().max(), min=1e-300)
        K = torch.where(close, (0.5 * (fp[:, None] + fp[None, :])).expand(d, d), (f[:, None] - f[None, :]) / torch.where(close, torch.ones_like(dl), dl))
        t1 = torch.empty((d, d), dtype=torch.float64, device=Vr.device)
        M = torch.empty((d, d), dtype=torch.float64, device=Vr.device)
        sp = _stream_ptr(Vr)
        h.acquire(sp)
        try:
            h.gemm(False, False, d, d, d, 1.0, Vr.data_ptr(), d, g64.data_ptr(), d, 0.0, t1.data_ptr(), d)      # V' G   (rows of Vr = eigenvectors)
            h.gemm(False, True, d, d, d, 1.0, t1.data_ptr(), d, Vr.data_ptr(), d, 0.0, M.data_ptr(), d)        # V' G V
        finally:
            h.release(sp)
        MK = (M * K).contiguous()
        h.acquire(sp)
        try:
            h.gemm(True, False, d, d, d, 1.0, Vr.data_ptr(), d, MK.data_ptr(), d, 0.0, t1.data_ptr(), d)       # V (M o K)
            h.gemm(False, False, d, d, d, 1.0, t1.data_ptr(), d, Vr.data_ptr(), d, 0.0, M.data_ptr(), d)       # ... V'
        finally:
            h.release(sp)
        return M.to(ctx.dtype), None


def _inv_sqrtm(A: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """``A^-1/2`` with eigenvalues clamped at ``eps`` (device Jacobi EVD).  Differentiable: when ``A`` requires a gradient
    the result is a node of the autograd graph (:class:`_InvSqrtmFn`); otherwise one fused call (``ccz_inv_sqrtm``)."""
    _require_cuda(A, "_inv_sqrtm")
    if A.requires_grad and torch.is_grad_enabled():
        return _InvSqrtmFn.apply(A, float(eps))
    h = _backend.handle_for([A])
    a64 = A.detach().to(torch.float64).contiguous()
    out = torch.empty_like(a64)
    sp = _stream_ptr(A)
    h.acquire(sp)
    h.check(h.lib.ccz_inv_sqrtm(h.raw, C.c_void_p(a64.data_ptr()), a64.shape[0], float(eps),
                                C.c_void_p(out.data_ptr())))
    h.release(sp)
    return out.to(A.dtype)


class CCALoss(nn.Module):
    r"""Andrew et al. (2013) deep-CCA loss ``-||S11^-1/2 S12 S22^-1/2||_F^2`` for two views.

    Args:
        eps: ridge added to the within-view batch covariances (default 1e-5).
    """

    def __init__(self, eps: float = 1e-5) -> None:
        super().__init__()
        self.eps = eps

    def forward(self, representations: list[torch.Tensor]) -> torch.Tensor:
        if len(representations) != 2:
            raise ValueError(
                "CCALoss expects exactly 2 representations, "
                f"got {len(representations)}."
            )
        z1, z2 = representations
        from cca_zoo_amd import _dist

        if _dist.is_sharded():
            return _ShardedCCALossFn.apply(z1, z2, self.eps)
        if z1.dim() != 2 or z2.dim() != 2 or z1.shape[0] != z2.shape[0]:
            raise ValueError("CCALoss expects two (batch, d_i) tensors with equal batch size")
        return _PairLossFn.apply(self.eps, "CCALoss", z1, z2)


class _ShardedPairLossFn(torch.autograd.Function):
    """``MCCALoss`` of a batch whose rows are spread over the ranks of ``row_sharded()``: K1 on the local rows of
    ``[z_1 .. z_m]``, one all-reduce of the packed moments, ``ccz_pair_loss_moments`` replicated (one Cholesky + inverse
    per view) and the gradient of the GLOBAL loss with respect to the LOCAL rows as one ``ccz_transform`` GEMM
    ``(Z - mean) Gamma``."""

    @staticmethod
    def forward(ctx, eps: float, *zs: torch.Tensor) -> torch.Tensor:
        from cca_zoo_amd import _dist

        for z in zs:
            _require_cuda(z, "MCCALoss")
            if z.dim() != 2 or z.shape[0] != zs[0].shape[0]:
                raise ValueError("MCCALoss expects (batch, d_i) tensors with equal batch size")
        dt = zs[0].dtype
        zcat = torch.cat([z.to(dt) for z in zs], dim=1).contiguous()
        n_local, D = int(zcat.shape[0]), int(zcat.shape[1])
        dims = [int(z.shape[1]) for z in zs]
        dev = zcat.device
        h = _backend.handle_for([zcat])
        mom = torch.empty(D * D + D, dtype=torch.float64, device=dev)
        sp = _stream_ptr(zcat)
        h.acquire(sp)
        h.moments([(zcat.data_ptr(), D, D)], n_local, _backend.F32 if dt == torch.float32 else _backend.F64, True, mom.data_ptr(),
                  pilot=True, timed=False)
        n_total = n_local
        if _dist.is_sharded():
            npk = D * (D + 1) // 2 + D
            packed = torch.empty(npk + 1, dtype=torch.float64, device=dev)
            h.moments_pack(mom.data_ptr(), D, packed.data_ptr())
            h.release(sp)
            packed[npk:].fill_(float(n_local))
            n_total = _dist.allreduce_moments(packed, _dist.active_group())
            h.acquire(sp)
            h.moments_unpack(packed.data_ptr(), D, mom.data_ptr())
        need = any(ctx.needs_input_grad[1:])
        loss_h = C.c_double(0.0)
        gam = torch.empty((D, D), dtype=torch.float64, device=dev) if need else None
        mean = torch.empty(D, dtype=torch.float64, device=dev) if need else None
        dims_a = (C.c_int64 * len(dims))(*dims)
        h.check(h.lib.ccz_pair_loss_moments(h.raw, C.c_void_p(mom.data_ptr()), int(n_total), dims_a, len(dims), float(eps),
                                            C.byref(loss_h), C.c_void_p(gam.data_ptr()) if need else None,
                                            C.c_void_p(mean.data_ptr()) if need else None))
        if need:
            ctx.save_for_backward(_project(zcat, mean, gam))
            ctx.dims = dims
            ctx.dtypes = [z.dtype for z in zs]
        return torch.tensor(loss_h.value, dtype=dt, device=dev)

    @staticmethod
    def backward(ctx, grad_out):
        (gcat,) = ctx.saved_tensors
        grads = torch.split(gcat, ctx.dims, dim=1)
        return (None, *[(grad_out * g).to(t) for g, t in zip(grads, ctx.dtypes)])


class MCCALoss(nn.Module):
    r"""Sum of pairwise :class:`CCALoss` over all view pairs ``i < j``.

    Args:
        eps: ridge passed to every pairwise loss (default 1e-5).
    """

    #: views one fused pass serves (``ccz_pair_loss``); more fall back to the pair-by-pair sum
    _MAX_FUSED_VIEWS = 8

    def __init__(self, eps: float = 1e-5) -> None:
        super().__init__()
        self.eps = eps
        self._cca_loss = CCALoss(eps=eps)

    def forward(self, representations: list[torch.Tensor]) -> torch.Tensor:
        from cca_zoo_amd import _dist

        n_views = len(representations)
        total = torch.tensor(0.0, device=representations[0].device)      # fp32 accumulator, as the reference (:149)
        if 2 <= n_views <= self._MAX_FUSED_VIEWS:
            if _dist.is_sharded():
                return total + _ShardedPairLossFn.apply(self.eps, *representations)
            return total + _PairLossFn.apply(self.eps, "MCCALoss", *representations)
        for i in range(n_views):
            for j in range(i + 1, n_views):
                total = total + self._cca_loss([representations[i], representations[j]])
        return total


def _project(x: torch.Tensor, mean64: torch.Tensor, w64: torch.Tensor) -> torch.Tensor:
    """``(x - mean) @ w`` through ``ccz_transform`` (x: n x d CUDA tensor, mean: d, w: d x k float64 CUDA)."""
    h = _backend.handle_for([x])
    a = _view_of(x)                                        # e.g. an expanded gradient
    out = torch.empty((a.shape[0], w64.shape[1]), dtype=a.dtype, device=a.device)
    sp = _stream_ptr(a)
    h.acquire(sp)
    h.check(h.lib.ccz_transform(h.raw, _backend.F32 if a.dtype == torch.float32 else _backend.F64,
                                C.c_void_p(a.data_ptr()), a.shape[0], a.shape[1], a.stride(0),
                                C.c_void_p(mean64.data_ptr()), C.c_void_p(w64.data_ptr()), w64.shape[1],
                                C.c_void_p(out.data_ptr()), out.stride(0)))
    h.release(sp)
    return out


class _GCCALossFn(torch.autograd.Function):
    """MAX-VAR GCCA loss from second moments (reference: cca_zoo/deep/objectives.py:155-220).

    The reference whitens every view with an eigen inverse square root, forms the n x n Gram of the stacked
    whitened views and sums its top-k eigenvalues.  The non-zero spectrum of that n x n matrix is
    ``(n-1) x`` the generalised eigenvalues of ``C u = lambda B u`` with ``C`` the centred covariance of the
    stacked views and ``B = blockdiag(C_ii) + eps I``: one K1 pass over ``[z_1 .. z_m]``, one D x D
    generalised top-k eigen-solve (``ccz_gevp_topk``), and the closed-form gradient
    ``dL/dZ = (Z - mean) Gamma``, ``Gamma = -2 sum_k (u u' - lambda blockdiag(u_i u_i'))`` as one MFMA GEMM.
    """

    @staticmethod
    def forward(ctx, eps: float, *zs: torch.Tensor) -> torch.Tensor:
        for z in zs:
            _require_cuda(z, "GCCALoss")
            if z.dim() != 2 or z.shape[0] != zs[0].shape[0]:
                raise ValueError("GCCALoss expects (batch, d_i) tensors with equal batch size")
        dt = zs[0].dtype
        zcat = torch.cat([z.to(dt) for z in zs], dim=1).contiguous()
        n, D = int(zcat.shape[0]), int(zcat.shape[1])
        dims = [int(z.shape[1]) for z in zs]
        k = dims[0]
        dev = zcat.device
        h = _backend.handle_for([zcat])
        mom = torch.empty(D * D + D, dtype=torch.float64, device=dev)
        h.acquire(_stream_ptr(zcat))
        h.moments([(zcat.data_ptr(), D, D)], n, _backend.F32 if dt == torch.float32 else _backend.F64, True, mom.data_ptr(),
                  pilot=True, timed=False)
        # C, B = blockdiag(C_ii) + eps I, the top-k generalised eigenpairs and Gamma are all built on the device
        # from the moments (ccz_gcca_loss_moments); only the k eigenvalues (as the loss) come back to the host
        need = any(ctx.needs_input_grad[1:])
        loss_h = C.c_double(0.0)
        gam = torch.empty((D, D), dtype=torch.float64, device=dev) if need else None
        mean = torch.empty(D, dtype=torch.float64, device=dev) if need else None
        dims_a = (C.c_int64 * len(dims))(*dims)
        h.check(h.lib.ccz_gcca_loss_moments(h.raw, C.c_void_p(mom.data_ptr()), n, dims_a, len(dims), float(eps), k,
                                            C.byref(loss_h), C.c_void_p(gam.data_ptr()) if need else None,
                                            C.c_void_p(mean.data_ptr()) if need else None))
        loss = torch.tensor(loss_h.value, dtype=dt, device=dev)
        if need:
            ctx.save_for_backward(_project(zcat, mean, gam))
            ctx.dims = dims
            ctx.dtypes = [z.dtype for z in zs]
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (gcat,) = ctx.saved_tensors
        grads = torch.split(gcat, ctx.dims, dim=1)
        return (None, *[(grad_out * g).to(t) for g, t in zip(grads, ctx.dtypes)])


class GCCALoss(nn.Module):
    r"""MAX-VAR generalised CCA loss for any number of views: ``-sum_{d<=k} lambda_d(sum_i H_i H_i')`` with
    ``H_i`` the centred, ridge-whitened representation of view i and k the width of the first view.

    Args:
        eps: ridge added to the within-view batch covariances (default 1e-5).
    """

    def __init__(self, eps: float = 1e-5) -> None:
        super().__init__()
        self.eps = eps

    def forward(self, representations: list[torch.Tensor]) -> torch.Tensor:
        return _GCCALossFn.apply(self.eps, *representations)
