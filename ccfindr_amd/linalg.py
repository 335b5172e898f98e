"""Truncated SVD of the count matrix on the MI355X engine: the ``irlba::irlba(mat, rank)`` of the reference's
``svd2`` initialiser (R/bayesian.R:150-159) with its two sparse products ``X V`` and ``t(X) U`` run by the sweep
machinery (``k_spmm``) and, by default, everything else on the device as well (``vbnmf_engine_svd``: the subspace
never leaves HBM inside the iteration, up to 64 columns; wider subspaces -- ranks above 54 -- take the host-QR form);
k = rank + oversampling <= 128 (VBNMF_MAX_RANK).

irlba is an implicitly restarted Lanczos bidiagonalisation with ``tol = 1e-5``; this is block subspace iteration with
the same kind of stopping rule (relative change of the leading singular values) and a tighter default, so the
triplets agree with a full SVD to well below irlba's own tolerance when the spectrum has a gap after ``rank``.
"""
from __future__ import annotations

import numpy as np

from . import _native as N
from .engine import CountMatrix, VBEngine


def truncated_svd(mat, rank, tol=1e-7, maxit=60, oversample=10, seed=0, device=0, method="device"):
    """Leading ``rank`` singular triplets ``(u, d, vt)`` of the count matrix (``u`` n x rank, ``vt`` rank x m).

    ``method="device"`` (default): the whole subspace iteration runs on the GPU (``vbnmf_engine_svd``: sparse products,
    CholeskyQR2, Jacobi eigen-solve of the k x k Gram matrix; the host reads k numbers per iteration from pinned
    memory).  ``method="host_qr"``: the products on the GPU, the k-column QR and the small SVD in numpy (round 1's
    form; also what the device form falls back to when X has fewer than k independent directions)."""
    M = mat if isinstance(mat, CountMatrix) else CountMatrix(mat)
    own = M is not mat
    n, m = M.shape
    rank = int(rank)
    if rank < 1 or rank > min(n, m):
        raise ValueError("rank must be in [1, min(nrow, ncol)]")
    k = int(min(max(rank + oversample, rank), N.MAX_RANK, n, m))
    if k < rank:
        raise ValueError(f"rank {rank} exceeds the engine's maximum of {N.MAX_RANK}")
    if k > N.MAX_SVD_COLUMNS:
        method = "host_qr"                   # the device-resident form holds at most 64 columns; the products still run on the GPU
    eng = VBEngine(M, k, device=device)
    try:
        if method == "device":
            try:
                u, d, vt, _ = eng.svd(rank, tol=tol, maxit=maxit, seed=seed)
                return u, d, vt
            except N.VBNMFError as exc:
                if exc.code != N.ERR_STATE:
                    raise                                                     # rank-deficient subspace: the QR form copes
        rng = np.random.default_rng(seed)
        Q, _ = np.linalg.qr(eng.spmm(rng.standard_normal((k, m))))        # range finder: X G
        s_old = None
        for _ in range(maxit):
            Z, _ = np.linalg.qr(eng.spmm(Q, transpose=True).T)            # t(X) Q, orthonormalised (m x k)
            Q, Rm = np.linalg.qr(eng.spmm(Z.T))                           # X Z = Q Rm
            s = np.linalg.svd(Rm, compute_uv=False)
            if s_old is not None and np.max(np.abs(s[:rank] - s_old[:rank])) <= tol * s[0]:
                break
            s_old = s
        B = eng.spmm(Q, transpose=True)                                   # t(Q) X, k x m
        ub, d, vt = np.linalg.svd(B, full_matrices=False)
        u = Q @ ub
        return u[:, :rank], d[:rank], vt[:rank]
    finally:
        eng.close()
        if own:
            M.close()
