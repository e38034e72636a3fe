"""Differentiable CCA objectives on the MI355X solver core."""

from cca_zoo_amd.deep.objectives import CCALoss, MCCALoss

__all__ = ["CCALoss", "MCCALoss"]
