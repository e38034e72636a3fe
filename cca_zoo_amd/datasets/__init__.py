"""Synthetic multiview data (measurement input generator)."""

from cca_zoo_amd.datasets._simulated import JointData

__all__ = ["JointData"]
