"""Small post-processing mirrors that sit right after the factorisation drivers."""
from __future__ import annotations

import numpy as np


def cluster_id(result, rank=2):
    """``cluster_id(object, rank)`` (reference R/utils.R:903-909): for every cell the 1-based index of its largest
    coefficient in the factorisation of the given rank (``apply(h, 2, which.max)``, first maximum on ties).
    ``result`` is a ``VBResult`` or ``MLResult`` (anything with ``ranks`` and ``coeff``)."""
    hits = [i for i, r in enumerate(result.ranks) if r == rank]
    if not hits:
        raise IndexError("subscript out of bounds")          # coeff(object)[ranks(object) == rank][[1]] on an empty list
    h = np.asarray(result.coeff[hits[0]])
    return np.argmax(h, axis=0) + 1
