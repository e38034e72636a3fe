"""Batch whitening layer of DCCA-NOI (reference: cca_zoo/deep/_dcca_noi.py:12-67).

Only ``_BatchWhiten`` -- the part of that file on the covariance / eigen hot path -- is provided; the
Lightning ``DCCA_NOI`` trainer class is outside SURVEY.md section 8.  Per training step: the uncentred
second moment ``x'x`` comes from K1 (``ccz_moments``), the running matrix is an exponential moving average,
its clamped inverse square root is the device Jacobi EVD (``ccz_inv_sqrtm``) and ``x @ w`` / its gradient
``g @ w'`` are ``ccz_transform`` GEMMs.  No gradient flows through ``w`` (as in the reference).
"""

from __future__ import annotations

import torch
import torch.nn as nn

from cca_zoo_amd import _backend
from cca_zoo_amd.deep.objectives import _inv_sqrtm, _project, _require_cuda


class _WhitenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w64):
        ctx.save_for_backward(w64)
        return _project(x, torch.zeros(x.shape[1], dtype=torch.float64, device=x.device), w64)

    @staticmethod
    def backward(ctx, g):
        (w64,) = ctx.saved_tensors
        return _project(g, torch.zeros(g.shape[1], dtype=torch.float64, device=g.device), w64.t().contiguous()), None


class _BatchWhiten(nn.Module):
    """Whitens a batch with the inverse square root of a running second-moment estimate (training mode only;
    identity in eval mode).

    Args:
        num_features: width of the input.
        momentum: weight of the current batch in the running estimate (default 0.1).
        eps: floor applied to the eigenvalues before the inverse square root (default 1e-5).
    """

    def __init__(self, num_features: int, momentum: float = 0.1, eps: float = 1e-5) -> None:
        super().__init__()
        self.num_features = num_features
        self.momentum = momentum
        self.eps = eps
        self.register_buffer("running_covar", torch.eye(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.training:
            return x
        _require_cuda(x, "_BatchWhiten")
        self.num_batches_tracked.add_(1)
        n, d = int(x.shape[0]), int(x.shape[1])
        a = x.detach()
        a = a if (a.stride(1) == 1 and a.stride(0) >= a.shape[1]) else a.contiguous()
        h = _backend.handle_for([x])
        mom = torch.empty(d * d + d, dtype=torch.float64, device=x.device)
        sp = int(torch.cuda.current_stream(x.device).cuda_stream)
        h.acquire(sp)                                      # enqueue-only: the stream hand-over happens on the device
        h.moments([(a.data_ptr(), d, a.stride(0))], n, _backend.F32 if a.dtype == torch.float32 else _backend.F64,
                  True, mom.data_ptr(), pilot=True, timed=False)
        h.moments_symmetrize(mom.data_ptr(), d)
        h.release(sp)
        batch_cov = (mom[: d * d].reshape(d, d) / n).to(self.running_covar.dtype)
        with torch.no_grad():
            self.running_covar.mul_(1.0 - self.momentum).add_(batch_cov * self.momentum)
            w = _inv_sqrtm(self.running_covar.to(x.device), self.eps)
        return _WhitenFn.apply(x, w.to(torch.float64).contiguous())
