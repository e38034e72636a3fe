"""``BaseDeep.score``'s arithmetic on representations that are already computed.

reference: cca_zoo/deep/_base.py:159-173 -- ``transform(loader)`` gathers the encoders' outputs on the host, then
``MCCA(latent_dimensions).fit(representations).score(representations)``.  Here the representations may stay where the
encoders left them (CUDA tensors): K1, the MCCA solve, the projection and the score's second K1 pass all run on the device.
"""

from __future__ import annotations

import numpy as np


def score_representations(representations, latent_dimensions: int) -> np.ndarray:
    """Average pairwise canonical correlations of ``representations`` (one ``(n, d_i)`` array or CUDA tensor per view)
    after a linear MCCA with ``latent_dimensions`` components -- shape ``(latent_dimensions,)``."""
    from cca_zoo_amd.linear import MCCA

    reps = list(representations)
    if len(reps) < 2:
        raise ValueError("score_representations needs at least two views")
    if any(getattr(r, "requires_grad", False) for r in reps):
        reps = [r.detach() for r in reps]
    return np.asarray(MCCA(latent_dimensions=int(latent_dimensions)).fit(reps).score(reps))
