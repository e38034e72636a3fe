"""cca_zoo_amd -- MI355X-native solver core for the CCA Gram / eigen hot path.

Drop-in for that path of ``cca_zoo``: ``cca_zoo_amd.linear.{CCA, rCCA, PLS, MCCA, GCCA}``
keep the scikit-learn estimator surface; ``cca_zoo_amd.deep.objectives.{CCALoss,
MCCALoss}`` keep the ``nn.Module`` objective contract; everything numerical runs in
``libccz`` (hand-written HIP for gfx950) through ``ctypes``.  No CPU fallback.
"""

from cca_zoo_amd._dist import row_sharded, shard_bounds


def k1_route(route=None, device=None):
    """Arithmetic route of float32 views through the Gram kernel K1 on ``device`` (default: the current one):
    ``"auto"`` (default: the split-bf16 route from 32768 rows on, where it is both faster and at least as accurate as the fp32
    kernel), ``"fp32"`` (always the fp32 matrix pipe: the reference's own precision in kind) or ``"bf16x2"`` (always the split
    route; also opts the large projections of ``transform`` into it).  Returns the previous setting; ``None`` only queries.
    See ``include/ccz.h: ccz_k1_route``; the environment variable ``CCZ_K1_ROUTE`` overrides ``"auto"``."""
    from cca_zoo_amd import _backend

    return _backend.default_handle(device).k1_route(route)


__all__ = ["row_sharded", "shard_bounds", "k1_route", "linear", "deep", "datasets"]
__version__ = "0.1.0"
