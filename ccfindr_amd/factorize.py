"""Host-side mirror of ccfindR's maximum-likelihood driver ``factorize()`` (reference R/factorize.R) on the MI355X
engine: the per-iteration step (``nmf_updateR`` :2-27 and ``likelihood`` :40-49) runs in libvbnmf_hip.so behind
``vbnmf_engine_ml_*`` (include/vbnmf.h); the loop, the stopping criteria and the per-rank quality measures are the
reference's, line by line.  No CPU fallback: without the HIP library every call raises.
"""
from __future__ import annotations

import ctypes
import os
import sys
from dataclasses import dataclass, field

import numpy as np

from . import _native as N
from .engine import CountMatrix, VBEngine, auto_batch

BATCH_MAX_RANK = 16          # ranks the batch kernels are built for (csrc/engine.hip: kBatchMaxPaddedRank)

# Above this many cell pairs the O(m^2) connectivity vector (R/factorize.R:51-60) is not formed: dispersion
# and cophenetic are NaN (the reference would try to allocate it).
MAX_PAIRS = 60_000_000


def nmf_update(x, w, h, prior=False, gamma_a=1.0, gamma_b=1.0):
    """Stateless ``nmf_updateR(x, w, h, n, m, r, prior, gamma.a, gamma.b)`` (R/factorize.R:2-27) ->
    ``{"ew": w, "eh": h}`` (:26), plus ``"lk"``: ``likelihood(x, w, h)`` of the updated pair (:40-49), which the
    reference's only caller evaluates right after (:195-196)."""
    L = N.load()
    w0, h0 = N.fcol(w), N.fcol(h)
    n, r = w0.shape
    m = h0.shape[1]
    if h0.shape != (r, m):
        raise ValueError("w and h have inconsistent shapes")
    w1 = np.empty((n, r), order="F")
    h1 = np.empty((r, m), order="F")
    lk = ctypes.c_double()
    if hasattr(x, "tocsc"):
        S = x.tocsc()
        if S.shape != (n, m):
            raise ValueError("x and w, h have inconsistent shapes")
        p = np.ascontiguousarray(S.indptr, dtype=np.int32)
        i = np.ascontiguousarray(S.indices, dtype=np.int32)
        v = np.ascontiguousarray(S.data, dtype=np.float64)
        N.check(L.vbnmf_ml_update_csc(n, m, r, p.ctypes.data_as(N.c_int32_p), i.ctypes.data_as(N.c_int32_p), N.dptr(v),
                                      N.dptr(w0), N.dptr(h0), int(bool(prior)), float(gamma_a), float(gamma_b),
                                      N.dptr(w1), N.dptr(h1), ctypes.byref(lk)))
    else:
        A = N.fcol(x)
        if A.shape != (n, m):
            raise ValueError("x and w, h have inconsistent shapes")
        N.check(L.vbnmf_ml_update_dense(n, m, r, N.dptr(A), N.dptr(w0), N.dptr(h0), int(bool(prior)),
                                        float(gamma_a), float(gamma_b), N.dptr(w1), N.dptr(h1), ctypes.byref(lk)))
    return {"ew": w1, "eh": h1, "lk": lk.value}


def likelihood(mat, w, h, device=0):
    """``likelihood(mat, w, h)`` (R/factorize.R:40-49) evaluated on the device."""
    M = mat if isinstance(mat, CountMatrix) else CountMatrix(mat)
    eng = VBEngine(M, np.asarray(w).shape[1], device=device)
    try:
        eng.ml_set_state(w, h)
        return eng.ml_likelihood()
    finally:
        eng.close()


def init(nrow, ncol, rank, rng):
    """``init(nrow, ncol, mat, rank)`` (R/factorize.R:30-38): uniform(0, 1) factors; ``rng`` is a numpy Generator
    (R's RNG stream cannot be reproduced without R)."""
    w = rng.uniform(size=(nrow, rank))
    h = rng.uniform(size=(rank, ncol))
    return {"ew": w, "eh": h}


def cluster_ids(h):
    """``which.max(h[, j])[1]`` for every cell j (R/factorize.R:55-56), 0-based."""
    return np.argmax(np.asarray(h), axis=0)


def connectivity(h):
    """``connectivity(h)`` (R/factorize.R:51-60): for every pair of cells, same arg-max component or not, in the
    order of ``t(cnn)[lower.tri(t(cnn))]`` (the condensed order of a ``dist`` object)."""
    cid = cluster_ids(h)
    m = cid.shape[0]
    iu = np.triu_indices(m, 1)          # pairs (a < b) row by row == lower triangle column by column
    return cid[iu[0]] == cid[iu[1]]


def connectivity_changes(cid_old, cid_new, rank):
    """``sum(cnn != cnn0)`` (R/factorize.R:201) without the O(m^2) vectors: a pair changes when it is together in
    exactly one of the two partitions; counts come from the contingency table of the two labelings."""
    table = np.zeros((rank, rank), dtype=np.int64)
    np.add.at(table, (cid_old, cid_new), 1)
    pairs = lambda c: int(np.sum(c * (c - 1) // 2))
    both = pairs(table)
    return pairs(table.sum(axis=1)) + pairs(table.sum(axis=0)) - 2 * both


def dispersion(cnn, nc):
    """``dispersion(cnn, nc)`` (R/factorize.R:62-67)."""
    con = float(np.sum((np.asarray(cnn, dtype=np.float64) - 0.5) ** 2))
    return 1.0 / nc + 8.0 * con / nc ** 2


def cophenet(conav, nc, method="average"):
    """``cophenet(conav, nc, method)`` (R/factorize.R:69-78): correlation between 1 - connectivity and the
    cophenetic distances of its hierarchical clustering (scipy in place of R's hclust / cophenetic / cor)."""
    from scipy.cluster.hierarchy import cophenet as sc_cophenet
    from scipy.cluster.hierarchy import linkage
    d = 1.0 - np.asarray(conav, dtype=np.float64)
    c, _ = sc_cophenet(linkage(d, method=method), d)
    return float(c)


@dataclass
class MLResult:
    """The slots factorize() fills in the reference's scNMFSet (R/factorize.R:303-318)."""
    ranks: list = field(default_factory=list)
    basis: list = field(default_factory=list)       # w of the best run, n x r per rank (averaged over nsmpl)
    coeff: list = field(default_factory=list)       # h of the best run, r x m per rank
    measure: dict = field(default_factory=dict)     # rank, likelihood, dispersion, cophenetic [, r_se, d_se, c_se]
    metadata: dict = field(default_factory=dict)    # store.connectivity: nrun, connectivity
    nsteps: list = field(default_factory=list)      # iterations of every run of the last sample, per rank (not in the reference)


def factorize(mat, ranks=2, nrun=20, randomize=False, nsmpl=1, verbose=2, progress_bar=True, Itmax=10000,
              ncnn_step=40, criterion="likelihood", linkage="average", Tol=1e-5, store_connectivity=False,
              seed=None, device=0, engine_factory=None, device_loop=True, batch=None):
    """Maximum-likelihood NMF of a count matrix on the MI355X engine; reference R/factorize.R:140-320.

    ``mat``: genes x cells counts (dense array, scipy sparse, or ``CountMatrix``).  ``seed`` seeds the numpy
    Generator behind ``init`` and the ``randomize`` permutations.  ``engine_factory(count_matrix, rank)`` replaces
    the engine constructor (the CPU tests of this loop pass a stand-in).  ``device_loop``: under
    ``criterion='likelihood'`` (and ``verbose < 3``) the inner loop (:194-213) runs on the device
    (``vbnmf_engine_ml_run``) instead of one call per iteration.  ``batch``: how many of a rank's ``nrun`` restarts
    (:181) are stepped by ONE launch (``engine.run_batch_ml``; 1: one at a time; None: up to 16 where the device loop runs,
    the rank is at most 16 and the matrix holds up to 2e7 stored entries) -- on a small matrix one loop cannot fill the GPU.
    The restarts draw their starts from the same stream in the same order either way.  Returns ``MLResult``.
    """
    del progress_bar
    if isinstance(mat, CountMatrix):
        M = mat
        host = None
    else:
        M = CountMatrix(mat)
        host = mat
        M.host = mat                    # what the matrix was built from (engine factories of the CPU tests read it)
    er, ec = M.empty_counts()
    if er > 0:
        raise ValueError("Input matrix contains empty rows")                       # :151
    if ec > 0:
        raise ValueError("Input matrix contains empty columns")                    # :152
    nrow, ncol = M.shape
    ranks = [int(r) for r in np.atleast_1d(ranks)]
    nrank = len(ranks)
    npair = ncol * (ncol - 1) // 2                                                 # :161
    pairs_ok = npair <= MAX_PAIRS
    if criterion not in ("likelihood", "connectivity"):
        raise ValueError("Unknown stopping criterion.")                            # :215
    if randomize and host is None:
        raise ValueError("randomize needs the host matrix, not a CountMatrix")
    rng = np.random.default_rng(seed)
    out = MLResult(ranks=ranks)
    rave, dave, coav = np.zeros(nrank), np.zeros(nrank), np.zeros(nrank)
    rste, dste, cste = np.full(nrank, np.nan), np.full(nrank, np.nan), np.full(nrank, np.nan)
    conav = None
    say = lambda *a: (print(*a), sys.stdout.flush())

    for irank, rank in enumerate(ranks):
        if verbose > 0:
            say(f"Rank {rank}")
        wsum = hsum = None
        rdat, ddat, cdat = [], [], []
        for ismpl in range(1, nsmpl + 1):
            conav = np.zeros(npair) if pairs_ok else None                          # :174
            if randomize:                                                          # :175-176: shuffle every column
                A = np.array(host.toarray() if hasattr(host, "toarray") else host, dtype=np.float64)
                A = np.apply_along_axis(rng.permutation, 0, A)
                Ms = CountMatrix(A)
                Ms.host = A
            else:
                Ms = M
            # the restarts of this rank stepped `nb` at a time by one launch (engine.run_batch_ml), or one engine for all of them
            nb = 1
            if (engine_factory is None and device_loop and criterion == "likelihood" and verbose < 3 and nrun > 1 and rank <= BATCH_MAX_RANK
                    and (batch is None or int(batch) > 1) and os.environ.get("VBNMF_NO_CONTROL_FOLD", "0") != "1"):
                nb = min(nrun, int(batch), 64) if batch is not None else auto_batch(Ms.nnz, nrun)
            elif batch is not None and int(batch) > 1:
                raise ValueError("batch > 1 needs nrun > 1, rank <= %d, criterion 'likelihood', the device loop and the library's own engines" % BATCH_MAX_RANK)
            if nb > 1:
                from .engine import batch_grid, run_batch_ml
                batch_engines = [VBEngine(Ms, rank, device=device, grid=batch_grid(nb)) for _ in range(nb)]
                eng = batch_engines[0]
            else:
                batch_engines = None
                eng = engine_factory(Ms, rank) if engine_factory else VBEngine(Ms, rank, device=device)
            rmax, wmax, hmax, disp, steps = -np.inf, None, None, np.nan, []
            ahead = {}                                                             # batched: results of runs already stepped
            try:
                for irun in range(1, nrun + 1):
                    if verbose >= 2:
                        say(f"Rnd.sample # {ismpl} , run # {irun} :" if randomize else f"Run # {irun} :")
                    if batch_engines is not None:
                        if irun not in ahead:                                      # this run opens a chunk: draw the chunk's starts in run
                            chunk = list(range(irun, min(nrun, irun + nb - 1) + 1))                 # order (:192), step them together
                            for slot, jrun in enumerate(chunk):
                                whj = init(nrow, ncol, rank, rng)
                                batch_engines[slot].ml_set_state(whj["ew"], whj["eh"])
                            for slot, (jrun, res) in enumerate(zip(chunk, run_batch_ml(batch_engines[:len(chunk)], Itmax=Itmax, Tol=Tol))):
                                ahead[jrun] = (batch_engines[slot], res)
                        eng, run = ahead.pop(irun)
                    else:
                        wh = init(nrow, ncol, rank, rng)                           # :192
                        eng.ml_set_state(wh["ew"], wh["eh"])
                    zstep, lkold, cid0, lk0, it = 0, -np.inf, None, np.nan, 0
                    on_device = device_loop and criterion == "likelihood" and verbose < 3 and hasattr(eng, "ml_run")
                    if batch_engines is not None:
                        it, lk0 = run["it"], run["lk"]
                    elif on_device:                                                # the loop below, driven by the device
                        run = eng.ml_run(Itmax=Itmax, Tol=Tol)
                        it, lk0 = run["it"], run["lk"]
                    for it in (() if on_device else range(1, Itmax + 1)):          # :196
                        lk0 = eng.ml_step()                                        # :197-198
                        if criterion == "connectivity":
                            if hasattr(eng, "cluster_changes"):            # labels, table and pair count on the device
                                nchange, cid = eng.cluster_changes()[0], None
                                if nchange is None or it == 1:
                                    nchange = npair                        # :200
                            else:
                                cid = (eng.cluster_ids() - 1) if hasattr(eng, "cluster_ids") else cluster_ids(eng.ml_get_state(("eh",))["eh"])
                                nchange = npair if it == 1 else connectivity_changes(cid0, cid, rank)   # :200-202
                            if verbose >= 3:
                                say(f"{it} : likelihood =  {lk0} , connectivity change =  {nchange}")
                            zstep = zstep + 1 if nchange == 0 else 0               # :206-207
                            if zstep == ncnn_step:
                                break                                              # :208
                            cid0 = cid
                        else:
                            if abs(lkold - lk0) < Tol * abs(lkold):                # :212
                                break
                            if verbose >= 3:
                                say(f"{it} : likelihood =  {lk0}")
                            lkold = lk0
                    steps.append(it)
                    st = eng.ml_get_state()
                    if pairs_ok:
                        conav = conav + connectivity(st["eh"])                     # :218-219
                        disp = dispersion(conav / irun, ncol)                      # :220
                    if verbose >= 2:
                        say(f"Nsteps = {it} , likelihood = {lk0} , dispersion = {disp}\n")
                    if (irun == 1 or lk0 > rmax) and not np.isnan(lk0):            # :223
                        rmax, wmax, hmax = lk0, st["ew"], st["eh"]
            finally:
                for be in (batch_engines if batch_engines is not None else [eng]):
                    be.close()
                if randomize:
                    Ms.close()
            coph = cophenet(conav / nrun, ncol, method=linkage) if pairs_ok else np.nan    # :230
            if verbose >= 1:
                say(f"Sample# {ismpl} : Max(likelihood) = {rmax} , dispersion = {disp} , cophenetic = {coph}")
            if wmax is None:
                raise FloatingPointError("every run of this rank ended with a NaN likelihood")
            wsum = wmax if wsum is None else wsum + wmax                           # :233-245
            hsum = hmax if hsum is None else hsum + hmax
            rdat.append(rmax); ddat.append(disp); cdat.append(coph)
        out.basis.append(wsum / nsmpl)                                             # :247-248
        out.coeff.append(hsum / nsmpl)
        out.nsteps.append(steps)
        if nsmpl > 1:                                                              # :249-254
            denom = np.sqrt(nsmpl - 1)
            rste[irank] = np.std(rdat, ddof=1) / denom
            dste[irank] = np.std(ddat, ddof=1) / denom
            cste[irank] = np.std(cdat, ddof=1) / denom
        rave[irank], dave[irank], coav[irank] = np.mean(rdat), np.mean(ddat), np.mean(cdat)   # :258-260
        if verbose >= 1 and randomize:
            say(f"Mean(likelihood) =  {rave[irank]} , Mean(dispersion) = {dave[irank]} , Mean(cophenetic) = {coav[irank]}\n")
    if randomize:                                                                  # :307-312
        out.measure = {"rank": list(ranks), "likelihood": rave.tolist(), "r_se": rste.tolist(),
                       "dispersion": dave.tolist(), "d_se": dste.tolist(), "cophenetic": coav.tolist(), "c_se": cste.tolist()}
    else:
        out.measure = {"rank": list(ranks), "likelihood": rave.tolist(), "dispersion": dave.tolist(),
                       "cophenetic": coav.tolist()}
    if store_connectivity and conav is not None:                                   # :315-316
        out.metadata = {"nrun": nrun, "connectivity": conav / nrun}
    if host is not None:
        M.close()
    return out
