"""ccfindr_amd -- MI355X-native engine for ccfindR's variational-Bayes NMF update path.

Only the hot path of the reference is here (reference src/vbnmf_update.cpp behind
R/bayesian.R's vb_factorize / vb_iterate); see DESIGN.md for scope.  The compute runs in
libvbnmf_hip.so (hand-written HIP for gfx950) behind the C ABI of include/vbnmf.h; there is
no CPU fallback.
"""
from ._native import LIB_PATH, MAX_RANK, VBNMFError, load  # noqa: F401
from .engine import EPS, Communicator, CountMatrix, VBEngine, batch_grid, run_batch, run_batch_ml  # noqa: F401
from .bayesian import (VBResult, hyper_update, vb_factorize, vb_init, vb_iterate,  # noqa: F401
                       vbnmf_update)

from .io import CountData, read_10x, remove_zeros, write_10x  # noqa: F401
from .post import cluster_id  # noqa: F401
from .factorize import MLResult, factorize, likelihood, nmf_update  # noqa: F401

__version__ = "0.1.0"
