"""Closed-form linear CCA family on the MI355X solver core."""

from cca_zoo_amd.linear._gcca import GCCA
from cca_zoo_amd.linear._grcca import GRCCA
from cca_zoo_amd.linear._mcca import MCCA
from cca_zoo_amd.linear._partialcca import PartialCCA
from cca_zoo_amd.linear._rcca import CCA, PLS, rCCA

__all__ = ["CCA", "GCCA", "GRCCA", "MCCA", "PLS", "PartialCCA", "rCCA"]
