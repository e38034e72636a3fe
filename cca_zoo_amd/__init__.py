"""cca_zoo_amd -- MI355X-native solver core for the CCA Gram / eigen hot path.

Drop-in for that path of ``cca_zoo``: ``cca_zoo_amd.linear.{CCA, rCCA, PLS, MCCA, GCCA}``
keep the scikit-learn estimator surface; ``cca_zoo_amd.deep.objectives.{CCALoss,
MCCALoss}`` keep the ``nn.Module`` objective contract; everything numerical runs in
``libccz`` (hand-written HIP for gfx950) through ``ctypes``.  No CPU fallback.
"""

from cca_zoo_amd._dist import row_sharded, shard_bounds

__all__ = ["row_sharded", "shard_bounds", "linear", "deep", "datasets"]
__version__ = "0.1.0"
