"""Oracle for the DCCA correlation loss.  TEST INFRASTRUCTURE ONLY.

``*_autograd`` functions restate ``cca_zoo/deep/objectives.py`` as the reference
structures it (CPU torch ``linalg.eigh`` + autograd); ``cca_loss_closed_form``
is the NumPy float64 specification of what the HIP path computes
(``loss = -tr(S11^-1 S12 S22^-1 S21)`` and its analytic gradient, SURVEY.md
section 8(a) rows 8-9).  Pinned by ``tests/golden/loss_*.npz``.
"""

from __future__ import annotations

import numpy as np
import torch

__all__ = [
    "inv_sqrtm_eigh",
    "cca_loss_autograd",
    "mcca_loss_autograd",
    "cca_loss_closed_form",
    "cca_loss_from_moments",
    "mcca_loss_closed_form",
]


def inv_sqrtm_eigh(A: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """objectives.py:9-21 -- eigh, clamp eigenvalues at eps, V diag(l^-1/2) V'."""
    lam, V = torch.linalg.eigh(A)
    lam = torch.clamp(lam, min=eps)
    return V @ torch.diag(lam.rsqrt()) @ V.T


def cca_loss_autograd(z1: torch.Tensor, z2: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """objectives.py:61-102 -- two-view loss exactly as the reference orders it."""
    n = z1.shape[0]
    a = z1 - z1.mean(dim=0)
    b = z2 - z2.mean(dim=0)
    eye1 = torch.eye(a.shape[1], dtype=a.dtype, device=a.device)
    eye2 = torch.eye(b.shape[1], dtype=b.dtype, device=b.device)
    s11 = a.T @ a / (n - 1) + eps * eye1
    s22 = b.T @ b / (n - 1) + eps * eye2
    s12 = a.T @ b / (n - 1)
    t = inv_sqrtm_eigh(s11, eps) @ s12 @ inv_sqrtm_eigh(s22, eps)
    ev = torch.linalg.eigvalsh(t.T @ t)
    return -torch.clamp(ev, min=0.0).sum()


def mcca_loss_autograd(zs, eps: float = 1e-5) -> torch.Tensor:
    """objectives.py:138-153 -- sum over i<j of the two-view loss (fp32 accumulator)."""
    total = torch.tensor(0.0, device=zs[0].device)
    for i in range(len(zs)):
        for j in range(i + 1, len(zs)):
            total = total + cca_loss_autograd(zs[i], zs[j], eps)
    return total


def cca_loss_closed_form(z1, z2, eps=1e-5):
    """float64 value and gradients from second moments + Cholesky solves.

    With a, b the centred inputs, ``S11 = a'a/(n-1) + eps I`` (same for S22),
    ``S12 = a'b/(n-1)``, ``P = S11^-1 S12``, ``Q = S22^-1 S12'``:

    * loss = -tr(P Q)
    * dL/dS12 = -2 S11^-1 S12 S22^-1 ;  dL/dS11 = P Q S11^-1 ;  dL/dS22 = Q P S22^-1
    * dz1 = center( (a (G11 + G11') + b G12') / (n-1) ), dz2 likewise.
    """
    z1 = np.asarray(z1, dtype=np.float64)
    z2 = np.asarray(z2, dtype=np.float64)
    n = z1.shape[0]
    a = z1 - z1.mean(axis=0)
    b = z2 - z2.mean(axis=0)
    s11 = a.T @ a / (n - 1) + eps * np.eye(a.shape[1])
    s22 = b.T @ b / (n - 1) + eps * np.eye(b.shape[1])
    s12 = a.T @ b / (n - 1)
    P = np.linalg.solve(s11, s12)            # d1 x d2
    Q = np.linalg.solve(s22, s12.T)          # d2 x d1
    loss = -np.trace(P @ Q)
    G12 = -2.0 * np.linalg.solve(s22, P.T).T       # -2 S11^-1 S12 S22^-1
    G11 = np.linalg.solve(s11, (P @ Q).T).T        # P Q S11^-1
    G22 = np.linalg.solve(s22, (Q @ P).T).T        # Q P S22^-1
    g1 = (a @ (G11 + G11.T) + b @ G12.T) / (n - 1)
    g2 = (b @ (G22 + G22.T) + a @ G12) / (n - 1)
    g1 -= g1.mean(axis=0)
    g2 -= g2.mean(axis=0)
    return loss, g1, g2


def cca_loss_from_moments(G, s, n, d1, d2, eps=1e-5):
    """The same closed form from the RAW second moments of the stacked batch ``[z1 | z2]`` -- ``G = Z'Z`` (D x D),
    ``s = 1'Z`` (D), ``n`` rows -- for batches too large to hold in float64 on the host (the metric shape,
    n = 1e6): returns ``(loss, Gamma, mean)`` with ``[dz1 | dz2] = (Z - 1 mean') Gamma`` for any subset of the rows.

    ``Gamma = [[G11 + G11', G12], [G12', G22 + G22']] / (n - 1)`` in the notation of :func:`cca_loss_closed_form`
    (cca_zoo/deep/objectives.py:61-102 + autograd); the column-centring of the gradients there is implied: columns of
    ``Z - 1 mean'`` sum to zero."""
    G = np.asarray(G, dtype=np.float64)
    s = np.asarray(s, dtype=np.float64)
    D = d1 + d2
    C = (G - np.outer(s, s) / n) / (n - 1)
    s11 = C[:d1, :d1] + eps * np.eye(d1)
    s22 = C[d1:, d1:] + eps * np.eye(d2)
    s12 = C[:d1, d1:]
    P = np.linalg.solve(s11, s12)
    Q = np.linalg.solve(s22, s12.T)
    loss = -float(np.sum(P * Q.T))                 # tr(P Q)
    G12 = -2.0 * np.linalg.solve(s22, P.T).T
    G11 = np.linalg.solve(s11, (P @ Q).T).T
    G22 = np.linalg.solve(s22, (Q @ P).T).T
    Gamma = np.empty((D, D))
    Gamma[:d1, :d1] = G11 + G11.T
    Gamma[:d1, d1:] = G12
    Gamma[d1:, :d1] = G12.T
    Gamma[d1:, d1:] = G22 + G22.T
    return loss, Gamma / (n - 1), s / n


def mcca_loss_closed_form(zs, eps=1e-5):
    """Sum of pairwise closed-form losses and the summed gradients."""
    zs = [np.asarray(z, dtype=np.float64) for z in zs]
    grads = [np.zeros_like(z) for z in zs]
    total = 0.0
    for i in range(len(zs)):
        for j in range(i + 1, len(zs)):
            l, gi, gj = cca_loss_closed_form(zs[i], zs[j], eps)
            total += l
            grads[i] += gi
            grads[j] += gj
    return total, grads


# --------------------------------------------------------------------------
# GCCALoss (cca_zoo/deep/objectives.py:155-220) and _BatchWhiten (deep/_dcca_noi.py:12-67)
# --------------------------------------------------------------------------
def gcca_loss_autograd(zs, eps: float = 1e-5) -> torch.Tensor:
    """Reference-structured: whiten every view with the eigen inverse square root, n x n Gram of the
    stacked whitened views, minus the sum of its top-k eigenvalues (k = width of the first view)."""
    n = zs[0].shape[0]
    hs = []
    for z in zs:
        zc = z - z.mean(dim=0)
        cov = zc.T @ zc / (n - 1) + eps * torch.eye(zc.shape[1], dtype=zc.dtype)
        hs.append(zc @ inv_sqrtm_eigh(cov, eps))
    m = sum(h @ h.T for h in hs)
    return -torch.linalg.eigvalsh(m)[-zs[0].shape[1]:].sum()


def gcca_loss_closed_form(zs, eps=1e-5):
    """Second-moment form (what the product computes).

    The non-zero spectrum of ``sum_i H_i H_i'`` (n x n) is that of ``[H_1..H_m]'[H_1..H_m]`` (D x D), which is
    ``(n-1) B^-1/2 C B^-1/2`` with ``C`` the centred covariance of the stacked views and
    ``B = blockdiag(C_ii) + eps I`` -- i.e. (n-1) times the generalised eigenvalues of ``C u = lambda B u``
    (any whitening of the blocks gives the same spectrum).  With ``u' B u = 1``:
    ``d lambda = u'(dC) u - lambda sum_i u_i'(dC_ii) u_i``, hence
    ``dL/dZ = Zc Gamma``,  ``Gamma = -2 sum_{top k} (u u' - lambda blockdiag(u_i u_i'))``.
    Returns (loss, [grad_i]) as float64 numpy arrays.
    """
    zs = [np.asarray(z, dtype=np.float64) for z in zs]
    n = zs[0].shape[0]
    dims = [z.shape[1] for z in zs]
    X = np.hstack(zs)
    Xc = X - X.mean(axis=0)
    C = Xc.T @ Xc / (n - 1)
    B = np.zeros_like(C)
    o = 0
    for d in dims:
        B[o:o + d, o:o + d] = C[o:o + d, o:o + d]
        o += d
    B += eps * np.eye(B.shape[0])
    import scipy.linalg

    lam, U = scipy.linalg.eigh(C, B)                       # ascending, U' B U = I
    k = dims[0]
    lam, U = lam[-k:], U[:, -k:]
    loss = -(n - 1) * lam.sum()
    Gamma = np.zeros_like(C)
    for j in range(k):
        u = U[:, j]
        Gamma += np.outer(u, u)
        o = 0
        for d in dims:
            Gamma[o:o + d, o:o + d] -= lam[j] * np.outer(u[o:o + d], u[o:o + d])
            o += d
    Gamma *= -2.0
    dX = Xc @ Gamma
    return loss, np.split(dX, np.cumsum(dims)[:-1], axis=1)


def batch_whiten_step(x, running, momentum, eps):
    """One training step of _BatchWhiten: EMA of the UNCENTRED second moment x'x/n, whitening by the eigen
    inverse square root (eigenvalues clamped at eps) of the updated running matrix; no gradient flows through
    the whitening matrix.  Returns (y, new_running, w) as float64 numpy arrays."""
    x = np.asarray(x, dtype=np.float64)
    running = (1.0 - momentum) * np.asarray(running, dtype=np.float64) + momentum * (x.T @ x / x.shape[0])
    vals, vecs = np.linalg.eigh(running)
    w = (vecs / np.sqrt(np.maximum(vals, eps))) @ vecs.T
    return x @ w, running, w
