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
"""

from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from cca_zoo_amd import _backend


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: expected a CUDA (ROCm) tensor; cca_zoo_amd runs on the MI355X only "
            "and has no CPU fallback"
        )
    if t.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"{what}: dtype must be float32 or float64, got {t.dtype}")


class _CCALossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z1: torch.Tensor, z2: torch.Tensor, eps: float) -> torch.Tensor:
        _require_cuda(z1, "CCALoss")
        _require_cuda(z2, "CCALoss")
        if z1.dtype != z2.dtype:
            z2 = z2.to(z1.dtype)
        if z1.dim() != 2 or z2.dim() != 2 or z1.shape[0] != z2.shape[0]:
            raise ValueError("CCALoss expects two (batch, d_i) tensors with equal batch size")
        a = z1 if z1.stride(1) == 1 else z1.contiguous()
        b = z2 if z2.stride(1) == 1 else z2.contiguous()
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        loss = torch.empty((), dtype=a.dtype, device=a.device)
        g1 = torch.empty_like(a, memory_format=torch.contiguous_format) if need else None
        g2 = torch.empty_like(b, memory_format=torch.contiguous_format) if need else None
        h = _backend.default_handle(a.device.index or 0)
        torch.cuda.current_stream(a.device).synchronize()
        h.check(h.lib.ccz_cca_loss(
            h.raw, _backend.F32 if a.dtype == torch.float32 else _backend.F64,
            C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), a.shape[0], a.shape[1], b.shape[1],
            a.stride(0), b.stride(0), float(eps), C.c_void_p(loss.data_ptr()),
            C.c_void_p(g1.data_ptr()) if need else None, C.c_void_p(g2.data_ptr()) if need else None,
            g1.stride(0) if need else 0, g2.stride(0) if need else 0))
        h.sync()
        if need:
            ctx.save_for_backward(g1, g2)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        g1, g2 = ctx.saved_tensors
        return grad_out * g1, grad_out * g2, None


def _inv_sqrtm(A: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """``A^-1/2`` with eigenvalues clamped at ``eps`` (device Jacobi EVD, forward only)."""
    _require_cuda(A, "_inv_sqrtm")
    h = _backend.default_handle(A.device.index or 0)
    a64 = A.detach().to(torch.float64).contiguous()
    out = torch.empty_like(a64)
    torch.cuda.current_stream(A.device).synchronize()
    h.check(h.lib.ccz_inv_sqrtm(h.raw, C.c_void_p(a64.data_ptr()), a64.shape[0], float(eps),
                                C.c_void_p(out.data_ptr())))
    h.sync()
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
        return _CCALossFn.apply(z1, z2, self.eps)


class MCCALoss(nn.Module):
    r"""Sum of pairwise :class:`CCALoss` over all view pairs ``i < j``.

    Args:
        eps: ridge passed to every pairwise loss (default 1e-5).
    """

    def __init__(self, eps: float = 1e-5) -> None:
        super().__init__()
        self.eps = eps
        self._cca_loss = CCALoss(eps=eps)

    def forward(self, representations: list[torch.Tensor]) -> torch.Tensor:
        n_views = len(representations)
        total = torch.tensor(0.0, device=representations[0].device)
        for i in range(n_views):
            for j in range(i + 1, n_views):
                total = total + self._cca_loss([representations[i], representations[j]])
        return total
