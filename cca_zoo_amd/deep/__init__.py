"""Differentiable CCA objectives on the MI355X solver core."""

from cca_zoo_amd.deep._score import score_representations
from cca_zoo_amd.deep.objectives import CCALoss, GCCALoss, MCCALoss

__all__ = ["CCALoss", "GCCALoss", "MCCALoss", "score_representations"]
