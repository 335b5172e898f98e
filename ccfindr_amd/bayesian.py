"""Host driver of the VB-NMF path: a Python mirror of the reference's R driver functions,
with the native step running on the MI355X engine instead of ``.Call(_ccfindR_vbnmf_update)``.

Mirrors (same names, argument meaning and error behaviour as the reference; R's dots in
argument names become underscores):

* ``vbnmf_update(X, wh, hyper, fudge)``   reference R/RcppExports.R:4-6 -> src/vbnmf_update.cpp:16-101
* ``hyper_update(...)``                    reference R/bayesian.R:2-53
* ``vb_init(...)``                         reference R/bayesian.R:109-171 (``random``, ``svd2``)
* ``vb_iterate(irun, bundle)``             reference R/bayesian.R:303-390
* ``vb_factorize(...)``                    reference R/bayesian.R:229-301

What is deliberately NOT here: the scNMFSet S4 container, connectivity/dispersion
(reference R/factorize.R:51-78, O(m^2) post-processing) and Rmpi.  ``vb_factorize`` takes the
count matrix itself and returns a plain result object holding the slots the reference fills.
"""
from __future__ import annotations

import math
import os
import threading
import warnings
from dataclasses import dataclass, field

import numpy as np

from . import _native as N
from .engine import EPS, CountMatrix, VBEngine, geometry_rank_for, rank_classes

import ctypes

BATCH_MAX_RANK = 16          # ranks the batch kernels are built for (csrc/engine.hip: kBatchMaxPaddedRank)


# ---------------------------------------------------------------------------------------
# scalar digamma / trigamma for the hyper-parameter Newton step (host side, a few calls
# per iteration): R's digamma() and psigamma(x, 1) at reference R/bayesian.R:19-24.
# ---------------------------------------------------------------------------------------
def _digamma(x: float) -> float:
    if not x > 0.0:
        return float("nan")
    s = 0.0
    while x < 10.0:
        s -= 1.0 / x
        x += 1.0
    xi = 1.0 / x
    y = xi * xi
    ser = y * (1 / 12 - y * (1 / 120 - y * (1 / 252 - y * (1 / 240 - y * (1 / 132 - y * (691 / 32760 - y / 12))))))
    return s + math.log(x) - 0.5 * xi - ser


def _trigamma(x: float) -> float:
    if not x > 0.0:
        return float("nan")
    s = 0.0
    while x < 10.0:
        s += 1.0 / (x * x)
        x += 1.0
    xi = 1.0 / x
    y = xi * xi
    ser = xi * y * (1 / 6 - y * (1 / 30 - y * (1 / 42 - y * (1 / 30 - y * (5 / 66 - y * (691 / 2730 - y * (7 / 6)))))))
    return s + xi + 0.5 * y + ser


# ---------------------------------------------------------------------------------------
def vbnmf_update(X, wh, hyper, fudge=EPS):
    """One update step, stateless: the reference's ``vbnmf_update(X, wh, hyper, fudge)``.

    ``X``: dense array (what ``as.matrix`` hands the reference) or scipy sparse matrix.
    ``wh``: mapping with ``lw`` (n x r), ``lh`` (r x m), ``eh`` (r x m) (``ew`` is accepted and
    ignored, as the reference overwrites it before use, src/vbnmf_update.cpp:24,44).
    ``hyper``: mapping with ``aw, bw, ah, bh``.  Returns the reference's list
    ``w, h, lw, lh, ew, eh, lkh, dw, dh`` (src/vbnmf_update.cpp:92-100) as a dict.
    """
    L = N.load()
    for key in ("lw", "lh", "eh"):
        if key not in wh:
            raise KeyError(f"wh has no member '{key}'")          # Rcpp: index out of bounds
    for key in ("aw", "bw", "ah", "bh"):
        if key not in hyper:
            raise KeyError(f"hyper has no member '{key}'")
    fud = float(np.asarray(fudge, dtype=np.float64).ravel()[0])  # fudge[0], src/vbnmf_update.cpp:19
    lw0, lh0, eh0 = N.fcol(wh["lw"]), N.fcol(wh["lh"]), N.fcol(wh["eh"])
    n, r = lw0.shape
    m = lh0.shape[1]
    if lh0.shape != (r, m) or eh0.shape != (r, m):
        raise ValueError("wh members have inconsistent shapes")
    lw = np.empty((n, r), order="F"); ew = np.empty((n, r), order="F"); dw = np.empty((n, r), order="F")
    lh = np.empty((r, m), order="F"); eh = np.empty((r, m), order="F"); dh = np.empty((r, m), order="F")
    lkh = ctypes.c_double()
    hy = [float(hyper[k]) for k in ("aw", "bw", "ah", "bh")]
    if hasattr(X, "tocsc"):
        S = X.tocsc()
        if S.shape != (n, m):
            raise ValueError("X and wh have inconsistent shapes")
        p = np.ascontiguousarray(S.indptr, dtype=np.int32)
        i = np.ascontiguousarray(S.indices, dtype=np.int32)
        x = np.ascontiguousarray(S.data, dtype=np.float64)
        N.check(L.vbnmf_update_csc(n, m, r, p.ctypes.data_as(N.c_int32_p), i.ctypes.data_as(N.c_int32_p), N.dptr(x),
                                   N.dptr(lw0), N.dptr(lh0), N.dptr(eh0), *hy, fud,
                                   N.dptr(lw), N.dptr(lh), N.dptr(ew), N.dptr(eh), N.dptr(dw), N.dptr(dh),
                                   ctypes.byref(lkh)))
    else:
        A = N.fcol(X)
        if A.shape != (n, m):
            raise ValueError("X and wh have inconsistent shapes")
        N.check(L.vbnmf_update_dense(n, m, r, N.dptr(A), N.dptr(lw0), N.dptr(lh0), N.dptr(eh0), *hy, fud,
                                     N.dptr(lw), N.dptr(lh), N.dptr(ew), N.dptr(eh), N.dptr(dw), N.dptr(dh),
                                     ctypes.byref(lkh)))
    return {"w": ew, "h": eh, "lw": lw, "lh": lh, "ew": ew, "eh": eh, "lkh": lkh.value, "dw": dw, "dh": dh}


def hyper_update(hyper_update, wh, hyper, Niter=100, Tol=1e-4):
    """Newton update of the Gamma shapes and the means; reference R/bayesian.R:2-53.

    ``wh`` is either the reference's list (``lw, lh, ew, eh`` matrices) or the 4-tuple of
    their reductions ``(mean log lw, mean log lh, mean ew, mean eh)`` the engine returns.
    """
    flags = [bool(f) for f in hyper_update]
    if sum(flags) == 0:                                          # :4
        return dict(hyper)
    aw0, ah0 = float(hyper["aw"]), float(hyper["ah"])
    if isinstance(wh, dict):
        lwm = float(np.mean(np.log(wh["lw"]))); lhm = float(np.mean(np.log(wh["lh"])))   # :8-9
        ewm = float(np.mean(wh["ew"])); ehm = float(np.mean(wh["eh"]))                   # :10-11
    else:
        lwm, lhm, ewm, ehm = (float(v) for v in wh)
    bw0, bh0 = float(hyper["bw"]), float(hyper["bh"])
    if flags[0] + flags[2] > 0:                                  # :15
        i = 1
        while i < Niter:                                         # :17
            dw = ((math.log(aw0) - _digamma(aw0) - ewm / bw0 + 1 + lwm - math.log(bw0))
                  / (1 / aw0 - _trigamma(aw0))) if flags[0] else 0.0
            dh = ((math.log(ah0) - _digamma(ah0) - ehm / bh0 + 1 + lhm - math.log(bh0))
                  / (1 / ah0 - _trigamma(ah0))) if flags[2] else 0.0
            # A non-finite Newton step (a -inf mean log: fudge = 0 and a shape small enough for exp(psi) to underflow)
            # makes the reference's halving loops below spin for ever; it is reported as the failure of :43 instead,
            # as the device loop does (kernels.h dev_hyper_update_pair).
            if not (math.isfinite(dw) and math.isfinite(dh)):
                raise RuntimeError("Hyper-parameter update failed to converge")
            aw1, ah1 = aw0 - dw, ah0 - dh
            while aw1 <= 0:                                      # :28-31
                dw /= 2
                aw1 = aw0 - dw
            while ah1 <= 0:                                      # :32-35
                dh /= 2
                ah1 = ah0 - dh
            df = (1 - aw1 / aw0) ** 2 + (1 - ah1 / ah0) ** 2     # :37
            if df < Tol:
                break
            aw0, ah0 = aw1, ah1
            i += 1
        if i == Niter:                                           # :43
            raise RuntimeError("Hyper-parameter update failed to converge")
    else:
        aw1, ah1 = aw0, ah0
    bw1 = ewm if flags[1] else bw0                               # :48-49
    bh1 = ehm                                                    # :50-51 (both branches assign ehm)
    return {"aw": aw1, "bw": bw1, "ah": ah1, "bh": bh1}


def vb_init(nrow, ncol, mat, rank, hyper, initializer, max=1.0, rng=None, device=0):
    """Initial ``wh``; reference R/bayesian.R:109-171.  ``random`` draws from the Gamma priors
    with a numpy Generator (R's RNG stream cannot be reproduced outside R); ``svd2`` takes
    |U| and |D V^T| of a rank-``rank`` SVD rescaled so mean(h) = bh -- the full SVD on the host for small
    matrices, as the reference does (:151-152), otherwise the truncated one on the device
    (``ccfindr_amd.linalg.truncated_svd`` in place of irlba, :154); ``svd`` is the NNDSVD-like start (:116-149), from
    the same leading triplets."""
    if initializer == "random":
        if rng is None:
            rng = np.random.default_rng()
        w = rng.gamma(shape=hyper["aw"], scale=hyper["bw"] / hyper["aw"], size=(nrow, rank))   # :112-113
        h = rng.gamma(shape=hyper["ah"], scale=hyper["bh"] / hyper["ah"], size=(rank, ncol))   # :114-115
    elif initializer == "svd2":
        if min(nrow, ncol) / 2 <= rank and not isinstance(mat, CountMatrix):
            A = mat.toarray() if hasattr(mat, "toarray") else np.asarray(mat, dtype=np.float64)
            u, d, vt = np.linalg.svd(A, full_matrices=False)                                    # :152
            u, d, vt = u[:, :rank], d[:rank], vt[:rank]
        else:
            from .linalg import truncated_svd                                                   # :154 (irlba), on the device
            u, d, vt = truncated_svd(mat, rank, seed=0 if rng is None else int(rng.integers(1 << 31)), device=device)
        w = np.abs(u)                                                                           # :155
        h = np.abs(np.diag(d) @ vt)                                                             # :156
        scale = hyper["bh"] / np.mean(h)                                                        # :157
        h = h * scale
        w = w / scale
    elif initializer == "svd":
        # NNDSVD-like start, reference R/bayesian.R:116-149, kept literally -- including :132-133, where the
        # norms of the NEGATIVE parts are computed from the positive parts (xp, yp), so mn == mp and the
        # positive branch (:135-138) is always taken; and :125, whose seq(2, rank) makes rank = 1 an error in R.
        if rank < 2:
            raise ValueError("initializer 'svd' needs rank >= 2 (reference R/bayesian.R:125 indexes component 2)")
        # :119 asks for nu = nv = rank and reads d[1..rank] only: the leading triplets are all it needs, so a large
        # (or already ingested) matrix goes through the device's truncated SVD instead of a dense full one, under
        # svd2's rule (:151-154); a dense array is decomposed on the host as the reference does.  The triplets' signs are
        # LAPACK's in the reference and the device's here: either is an arbitrary choice the formulas below depend on.
        stored_sparse = isinstance(mat, CountMatrix) or hasattr(mat, "toarray")
        if not stored_sparse or (min(nrow, ncol) / 2 <= rank and not isinstance(mat, CountMatrix)):
            A = mat.toarray() if hasattr(mat, "toarray") else np.asarray(mat, dtype=np.float64)
            u, d, vt = np.linalg.svd(A, full_matrices=False)                                    # :119
        else:
            from .linalg import truncated_svd
            u, d, vt = truncated_svd(mat, rank, seed=0 if rng is None else int(rng.integers(1 << 31)), device=device)
        w = np.zeros((nrow, rank)); h = np.zeros((rank, ncol))                                  # :117-118
        d1 = np.sqrt(d[0])                                                                      # :120
        w[:, 0] = d1 * u[:, 0]                                                                  # :121
        sgn = np.sign(w[0, 0])                                                                  # :122
        if sgn < 0:
            w = -w                                                                              # :123
        h[0, :] = sgn * d1 * vt[0, :]                                                           # :124
        for k in range(1, rank):                                                                # :125
            x, y = u[:, k], vt[k, :]
            xp, yp = np.where(x > 0, x, 0.0), np.where(y > 0, y, 0.0)                           # :128-129
            xn, yn = np.where(x < 0, -x, 0.0), np.where(y < 0, -y, 0.0)                         # :130-131
            xpnrm, ypnrm = np.sqrt(np.sum(xp ** 2)), np.sqrt(np.sum(yp ** 2))                   # :132-133
            mp = xpnrm * ypnrm
            xnnrm, ynnrm = np.sqrt(np.sum(xp ** 2)), np.sqrt(np.sum(yp ** 2))                   # :135-136 (sic: xp, yp)
            mn = xnnrm * ynnrm
            if mp >= mn:                                                                        # :138
                uu, vv, sig = xp / xpnrm, yp / ypnrm, mp
            else:
                uu, vv, sig = xn / xnnrm, yn / ynnrm, mn
            w[:, k] = np.sqrt(d[k] * sig) * uu                                                  # :147
            h[k, :] = np.sqrt(d[k] * sig) * vv                                                  # :148
    else:
        raise ValueError("Unknown initializer")                                                 # :160
    dw = np.zeros((nrow, rank)); dh = np.zeros((rank, ncol))
    return {"w": w, "h": h, "lw": w.copy(), "lh": h.copy(), "ew": w.copy(), "eh": h.copy(), "dw": dw, "dh": dh}


@dataclass
class VBResult:
    """The slots vb_factorize fills in the reference's scNMFSet (reference R/bayesian.R:293-299)."""
    ranks: list = field(default_factory=list)
    basis: list = field(default_factory=list)      # E[W], n x r per rank
    dbasis: list = field(default_factory=list)     # sd[W]
    coeff: list = field(default_factory=list)      # E[H], r x m per rank
    dcoeff: list = field(default_factory=list)     # sd[H]
    measure: dict = field(default_factory=dict)    # columns rank, lml, aw, bw, ah, bh, nunif
    nsteps: list = field(default_factory=list)     # iterations used by the selected run (not in the reference)


def _bundle_rng(bundle, irun, rank):
    """One independent stream per (run, rank): results do not depend on which process runs the task."""
    seed = bundle.get("seed")
    return np.random.default_rng(None if seed is None else [int(seed), int(irun), int(rank)])


def _engine_key(bundle, rank, slot=None):
    # under `concurrent` > 1 each worker thread keeps its own engines: an engine serves one host thread at a time;
    # a batch of restarts (vb_run_rank_batch) keeps one engine per slot of the batch
    if slot is not None:
        return ("slot", slot, rank)
    return (threading.get_ident(), rank) if bundle.get("concurrent", 1) > 1 else rank


def _make_engine(bundle, rank, slot=None):
    """The engine of one rank.  Building one means cutting the tiled layout of X for that rank on the host (seconds
    at C3), so the restarts of a rank (``nrun`` > 1) share it: ``bundle["engines"]`` keeps one per rank until
    ``_close_engines``."""
    cache = bundle.get("engines")
    key = _engine_key(bundle, rank, slot)
    if cache is not None and key in cache:
        return cache[key]
    factory = bundle.get("engine_factory")
    eng = (factory(bundle["mat"], rank) if factory is not None else
           VBEngine(bundle["mat"], rank, device=bundle.get("device", 0), geometry_rank=geometry_rank_for(rank, bundle.get("classes")),
                    grid=bundle.get("grid"), pad_rank=bundle.get("pad_rank")))
    if cache is not None:
        cache[key] = eng
    return eng


def _close_engines(bundle):
    cache = bundle.get("engines")
    if cache:
        for eng in cache.values():
            eng.close()
        cache.clear()


def vb_run_rank(irun, rank, bundle):
    """One factorisation (one run, one rank): the body of the rank loop, reference R/bayesian.R:318-384,
    with the engine as the update.  Returns the per-rank record vb_iterate stores."""
    X = bundle["mat"]
    nrow, ncol = X.shape
    verbose = bundle["verbose"]
    if rank > min(nrow, ncol):
        raise ValueError("Rank exceeded min(nrow,ncol)")                         # :319-320
    ga, gb = np.atleast_1d(bundle["gamma_a"]), np.atleast_1d(bundle["gamma_b"])
    hyper = {"aw": float(ga[0]), "ah": float(ga[-1]), "bw": float(gb[0]), "bh": float(gb[-1])}   # :321-326
    rng = _bundle_rng(bundle, irun, rank)
    raw = bundle.get("raw")
    clock = bundle.get("unit_times")                     # optional: seconds per phase of this unit (drivers' diagnostics)
    import time
    t_0 = time.perf_counter()
    eng = _make_engine(bundle, rank)
    t_eng = time.perf_counter()
    on_device = bundle.get("device_init") and bundle["initializer"] == "random" and hasattr(eng, "random_state")
    wh0 = None if on_device else vb_init(nrow, ncol, raw if raw is not None else X, rank, hyper=hyper,
                                         initializer=bundle["initializer"], rng=rng, device=bundle.get("device", 0))
    t_draw = time.perf_counter()
    try:
        if on_device:
            eng.random_state(hyper, int(rng.integers(1 << 63)))                  # :111-115 on the GPU
        else:
            eng.set_state(wh0["lw"], wh0["lh"], wh0["eh"])
        t_state = time.perf_counter()
        lk0 = 0.0
        it = 0
        device_loop = bundle.get("device_loop", True) and verbose < 3 and hasattr(eng, "run")
        if device_loop:
            # the whole loop below, driven by the device (same rules, no host round trip per step)
            out = eng.run(hyper, Itmax=bundle["Itmax"], Tol=bundle["Tol"], n0=bundle["hyper_update_n0"],
                          dn=bundle["hyper_update_dn"], flags=bundle["hyper_update"], fudge=bundle["fudge"])
            it, lk0, hyper = out["it"], out["lk0"], out["hyper"]
        for it in (() if device_loop else range(1, bundle["Itmax"] + 1)):       # :337
            lkh, stats = eng.step(hyper, bundle["fudge"])                        # :339
            if it > bundle["hyper_update_n0"] and it % bundle["hyper_update_dn"] == 0:   # :342
                hyper = hyper_update(bundle["hyper_update"], stats, hyper, Niter=100, Tol=1e-3)
            if math.isnan(lkh):                                                  # :345
                break
            if it > 1 and it > bundle["hyper_update_n0"]:                        # :346-347
                if lkh >= lk0 and abs(1 - lkh / lk0) < bundle["Tol"]:
                    break
            lk0 = lkh                                                            # :348
            if verbose >= 3:
                print(f"{it}, log(evidence) = {lk0}, aw = {hyper['aw']}, bw = {hyper['bw']}, "
                      f"ah = {hyper['ah']}, bh = {hyper['bh']}")
        # the unit's factor matrices: into caller-provided storage when the driver has some (a sharded sweep collects them
        # in memory shared by the node's processes, ccfindr_amd.parallel), else fresh arrays
        t_loop = time.perf_counter()
        slots = bundle["state_out"](irun, rank) if bundle.get("state_out") is not None else None
        if slots is not None and getattr(eng, "supports_state_out", False):
            wh = eng.get_state(("ew", "eh", "dw", "dh"), out=slots)
        else:
            wh = eng.get_state(("ew", "eh", "dw", "dh"))
            if slots is not None:
                for key in ("ew", "eh", "dw", "dh"):
                    slots[key][...] = wh[key]
                wh = slots
        if clock is not None:
            clock.append({"rank": rank, "engine_s": t_eng - t_0, "draw_s": t_draw - t_eng, "set_state_s": t_state - t_draw,
                          "loop_s": t_loop - t_state, "get_state_s": time.perf_counter() - t_loop, "it": it})
    finally:
        if bundle.get("engines") is None:
            eng.close()
    if verbose >= 2:
        print(f"Rank = {rank}: Nsteps ={it}, log(evidence) ={lk0}, hyper = ({hyper['aw']},{hyper['bw']},"
              f"{hyper['ah']},{hyper['bh']})")
    return _unit_record(rank, bundle, wh, slots is not None, lk0, hyper, it)


def _unit_record(rank, bundle, wh, in_place, lk0, hyper, it):
    """The per-rank record vb_iterate stores (reference R/bayesian.R:368-369, 382-383) from a finished unit's state."""
    ew = wh["ew"]
    contains_unif = np.abs(ew.max(axis=0) - ew.min(axis=0)) < bundle["Tol"]      # :368-369
    if in_place:                                                                 # sqrt in place: the storage is the record
        np.sqrt(wh["dw"], out=wh["dw"]); np.sqrt(wh["dh"], out=wh["dh"])
        sdw, sdh = wh["dw"], wh["dh"]
    else:
        sdw, sdh = np.sqrt(wh["dw"]), np.sqrt(wh["dh"])
    return {"rank": rank, "lk0": lk0, "ew": wh["ew"], "eh": wh["eh"], "sdw": sdw, "sdh": sdh,  # :382-383
            "hyper": hyper, "nsteps": it, "unif": [int(c) + 1 for c in np.nonzero(contains_unif)[0]]}


def vb_run_rank_batch(iruns, rank, bundle):
    """The restarts ``iruns`` of ONE rank, stepped together: ``vb_run_units_batch`` on the units (irun, rank)."""
    return vb_run_units_batch([(irun, rank) for irun in iruns], bundle)


def vb_run_units_batch(units, bundle):
    """The (run, rank) units ``units`` -- the restarts of a rank (reference R/bayesian.R:260-261: ``lapply(seq_len(nrun),
    vb_iterate)``), and, where the engines are made one row width wide (``bundle["pad_rank"]``), several ranks of the rank loop
    (:316) as well -- stepped together by ``engine.run_batch``: on a small matrix one loop cannot fill the GPU and concurrent
    streams do not overlap (profiles/r05_small_concurrent.txt), so the independent loops share their launches.  Every unit draws
    its start from its own (run, rank) stream and follows its own control block: the records are vb_run_rank's on engines of the
    same grid and width, bit for bit."""
    from .engine import run_batch
    X = bundle["mat"]
    nrow, ncol = X.shape
    ga, gb = np.atleast_1d(bundle["gamma_a"]), np.atleast_1d(bundle["gamma_b"])
    raw = bundle.get("raw")
    engines, hypers = [], []
    for slot, (irun, rank) in enumerate(units):
        if rank > min(nrow, ncol):
            raise ValueError("Rank exceeded min(nrow,ncol)")                     # :319-320
        hyper = {"aw": float(ga[0]), "ah": float(ga[-1]), "bw": float(gb[0]), "bh": float(gb[-1])}   # :321-326
        rng = _bundle_rng(bundle, irun, rank)
        eng = _make_engine(bundle, rank, slot)
        on_device = bundle.get("device_init") and bundle["initializer"] == "random" and hasattr(eng, "random_state")
        if on_device:
            eng.random_state(hyper, int(rng.integers(1 << 63)))                  # :111-115 on the GPU
        else:
            wh0 = vb_init(nrow, ncol, raw if raw is not None else X, rank, hyper=hyper,
                          initializer=bundle["initializer"], rng=rng, device=bundle.get("device", 0))
            eng.set_state(wh0["lw"], wh0["lh"], wh0["eh"])
        engines.append(eng); hypers.append(hyper)
    outs = run_batch(engines, hypers, Itmax=bundle["Itmax"], Tol=bundle["Tol"], n0=bundle["hyper_update_n0"],
                     dn=bundle["hyper_update_dn"], flags=bundle["hyper_update"], fudge=bundle["fudge"])
    if any(o["reason"] == 3 for o in outs):
        raise RuntimeError("Hyper-parameter update failed to converge")          # reference R/bayesian.R:43
    recs = []
    for (irun, rank), eng, out in zip(units, engines, outs):
        slots = bundle["state_out"](irun, rank) if bundle.get("state_out") is not None else None
        if slots is not None and getattr(eng, "supports_state_out", False):
            wh = eng.get_state(("ew", "eh", "dw", "dh"), out=slots)
        else:
            wh = eng.get_state(("ew", "eh", "dw", "dh"))
            if slots is not None:
                for key in ("ew", "eh", "dw", "dh"):
                    slots[key][...] = wh[key]
                wh = slots
        if bundle["verbose"] >= 2:
            hy = out["hyper"]
            print(f"Run {irun} rank = {rank}: Nsteps ={out['it']}, log(evidence) ={out['lk0']}, hyper = ({hy['aw']},{hy['bw']},"
                  f"{hy['ah']},{hy['bh']})")
        recs.append(_unit_record(rank, bundle, wh, slots is not None, out["lk0"], out["hyper"], out["it"]))
    return recs


def batch_across_ranks(bundle, across):
    """Whether a batch also takes several RANKS of the rank loop (reference R/bayesian.R:316): the engines of all ranks are then
    made one row width wide (the widest rank's), which costs the narrow ranks idle columns in the sweep -- nothing on a matrix
    whose step is latency bound anyway.  ``across`` as given, or -- ``None`` -- up to 2e6 stored entries."""
    if len(bundle["ranks"]) < 2 or across is False or len(bundle.get("classes") or []) > 1:      # (several geometry classes: several
        return False                                                                              # pairs of layouts, no common batch)
    return True if across else bundle["mat"].nnz <= 2_000_000


def batch_eligible(bundle, batch, across=None):
    """How many (run, rank) units vb_factorize steps together: ``batch`` as given (1: never), or -- ``None`` -- up to 16
    (``engine.auto_batch``: 16 / 8 / 4 by matrix size) where it pays and is possible: several units -- the ``nrun`` restarts of a
    rank, times the ranks where a batch spans ranks (``batch_across_ranks``) --, a matrix small enough that one loop leaves the
    GPU room (up to 2e7 stored entries: x 6 at 3.5e5, x 3.2 at 2e6, x 1.9 at 1.5e7 for eight restarts,
    profiles/r05_batch_sizes.txt), ranks within the batch kernels' range, the device-driven loop, the library's own engines."""
    if batch is not None and int(batch) <= 1:
        return 1
    units = bundle["nrun"] * (len(bundle["ranks"]) if batch_across_ranks(bundle, across) else 1)
    ok = (units > 1 and bundle.get("engine_factory") is None and bundle.get("device_loop", True) and bundle["verbose"] < 3 and
          bundle.get("concurrent", 1) == 1 and not getattr(bundle["mat"], "is_shell", False) and
          max(bundle["ranks"], default=0) <= BATCH_MAX_RANK and os.environ.get("VBNMF_NO_UPDATE_PAIR", "0") != "1" and
          os.environ.get("VBNMF_NO_CONTROL_FOLD", "0") != "1")
    if not ok:
        if batch is not None:
            raise ValueError("batch > 1 needs several (run, rank) units, ranks <= %d, the device-driven loop and the library's own engines" % BATCH_MAX_RANK)
        return 1
    if batch is None:
        from .engine import auto_batch
        return auto_batch(bundle["mat"].nnz, units)
    return min(int(batch), units, 64)


def vb_iterate_batched(bundle, batch):
    """All runs, the (run, rank) units stepped ``batch`` at a time (vb_run_units_batch).  With ``bundle["pad_rank"]`` set the
    engines of all ranks are one row width wide and a batch takes the runs still scanning times as many CONSECUTIVE ranks as fit
    (the higher ranks of a group are run ahead of the scan: a run whose scan ends inside the group -- a constant basis column
    under ``unif_stop``, reference R/bayesian.R:373-377 -- has their results dropped, exactly as if they had not been run);
    without it, rank by rank.  The records and the bookkeeping are vb_iterate's."""
    ranks = [int(r) for r in bundle["ranks"]]
    alive = list(range(1, bundle["nrun"] + 1))
    records = {irun: {} for irun in alive}
    across = bundle.get("pad_rank") is not None
    k = 0
    while k < len(ranks) and alive:
        width = max(1, batch // len(alive)) if across else 1                     # consecutive ranks taken together
        group = ranks[k:k + width]
        units = [(irun, rank) for rank in group for irun in alive]
        got = {}
        for c0 in range(0, len(units), batch):
            chunk = units[c0:c0 + batch]
            for unit, rec in zip(chunk, vb_run_units_batch(chunk, bundle)):
                got[unit] = rec
        for rank in group:                                                       # the scan, rank by rank (:316, :373-377)
            for irun in alive:
                records[irun][rank] = got[(irun, rank)]
            if bundle["unif_stop"]:
                alive = [irun for irun in alive if not records[irun][rank]["unif"]]
        cache = bundle.get("engines")                                            # these ranks' engines are done with
        for key in [q for q in (cache or {}) if isinstance(q, tuple) and q[0] == "slot" and q[2] in group]:
            cache.pop(key).close()
        k += len(group)
    return [assemble_run(records[irun], ranks, bundle["unif_stop"]) for irun in range(1, bundle["nrun"] + 1)]


def assemble_run(records, ranks, unif_stop):
    """The bookkeeping of one run over its rank records (reference R/bayesian.R:309-311, 368-388):
    a rank with a constant basis column warns; with unif.stop it ends the rank scan there."""
    nrank = len(ranks)
    out = {"rdat": [-math.inf] * nrank, "wdat": {}, "hdat": {}, "hyperp": {}, "nunif": [0] * nrank,
           "dwdat": {}, "dhdat": {}, "nsteps": {}}
    for irank, rank in enumerate(ranks):
        rec = records.get(rank)
        if rec is None:
            break
        if rec["unif"]:
            warnings.warn(f"Rank {rank} row/column {','.join(str(c) for c in rec['unif'])} constant.")
            if unif_stop:
                warnings.warn(f"Rank scan stopped for rank >= {rank}")
                if irank == 0:
                    raise RuntimeError("Rerun with lower ranks")                 # :375
                break
        out["rdat"][irank] = rec["lk0"]                                          # :379
        out["wdat"][irank] = rec["ew"]; out["hdat"][irank] = rec["eh"]
        out["dwdat"][irank] = rec["sdw"]; out["dhdat"][irank] = rec["sdh"]
        out["hyperp"][irank] = rec["hyper"]; out["nsteps"][irank] = rec["nsteps"]
    return out


def vb_iterate(irun, bundle):
    """One run over all ranks; reference R/bayesian.R:303-390."""
    ranks = [int(r) for r in bundle["ranks"]]
    if bundle["verbose"] >= 2 and bundle["nrun"] > 1:
        print(f"Run {irun}")
    records = {}
    for rank in ranks:
        rec = vb_run_rank(irun, rank, bundle)
        records[rank] = rec
        if rec["unif"] and bundle["unif_stop"]:
            break                                                                # :373-377
    return assemble_run(records, ranks, bundle["unif_stop"])


def select_best(vb, ranks):
    """Best run per rank by maximum log evidence; reference R/bayesian.R:265-299."""
    res = VBResult()
    cols = {k: [] for k in ("rank", "lml", "aw", "bw", "ah", "bh", "nunif")}
    for k, rank in enumerate(ranks):
        rmax, imax = -math.inf, None
        for i, run in enumerate(vb):
            if run["rdat"][k] > rmax:                                            # :271
                imax, rmax = i, run["rdat"][k]
        if rmax == -math.inf:                                                    # :276
            continue
        run = vb[imax]
        res.ranks.append(rank)
        res.basis.append(run["wdat"][k]); res.coeff.append(run["hdat"][k])
        res.dbasis.append(run["dwdat"][k]); res.dcoeff.append(run["dhdat"][k])
        res.nsteps.append(run["nsteps"][k])
        hy = run["hyperp"][k]
        for key, val in (("rank", rank), ("lml", rmax), ("aw", hy["aw"]), ("bw", hy["bw"]),
                         ("ah", hy["ah"]), ("bh", hy["bh"]), ("nunif", run["nunif"][k])):
            cols[key].append(val)
    res.measure = cols
    return res


def make_bundle(mat, ranks, nrun, verbose, initializer, Itmax, hyper_update, gamma_a, gamma_b, Tol,
                hyper_update_n0, hyper_update_dn, fudge, unif_stop, seed, device, engine_factory=None, check_empty=True):
    """Argument handling and guards of reference R/bayesian.R:238-259."""
    if fudge is None:
        fudge = EPS                                                              # :238
    if initializer in ("svd", "svd2") and nrun > 1:
        raise ValueError("SVD initializer does not require nrun > 1")            # :241-242
    X = mat if isinstance(mat, CountMatrix) else CountMatrix(mat)
    # (a shell -- CountMatrix.shell, a sharded sweep's processes that do not hold X -- has no entries to check: the
    # process that holds them ran these guards and every process raises on its findings, ccfindr_amd.parallel)
    nullr, nullc = (0, 0) if (getattr(X, "is_shell", False) or not check_empty) else X.empty_counts()   # :244-245
    if nullr > 0:
        raise ValueError("Input matrix contains empty rows")
    if nullc > 0:
        raise ValueError("Input matrix contains empty columns")
    ranks = [int(r) for r in np.atleast_1d(ranks) if r <= X.shape[1]]            # :249
    return {"mat": X, "raw": None if isinstance(mat, CountMatrix) else mat, "ranks": ranks, "verbose": verbose,
            "gamma_a": gamma_a, "gamma_b": gamma_b, "initializer": initializer, "Itmax": Itmax, "fudge": fudge,
            "hyper_update": list(hyper_update), "hyper_update_n0": hyper_update_n0,
            "hyper_update_dn": hyper_update_dn, "Tol": Tol, "unif_stop": unif_stop, "nrun": nrun,
            "seed": seed, "device": device, "engine_factory": engine_factory}


def plan_geometry(bundle, geometry_classes=1):
    """A sweep over several ranks shares ONE pair of tiled layouts: the geometry of the largest rank
    (``geometry_classes`` = 1), or up to that many classes (``ccfindr_amd.engine.rank_classes``).  Cutting a pair per LDS
    row size cost the reference-default sweep of BASELINE config C4 (ranks 2..20) 5 s of host time against 0.3 s of
    stepping; the price is a somewhat slower step at the lower ranks (narrower LDS blocks than their rows would allow).
    0 = every rank its own geometry.  The classes go into ``bundle["classes"]`` and from there to each engine
    (``VBEngine(geometry_rank=...)``): the matrix handle is not touched, so a plan the caller set with
    ``CountMatrix.plan_ranks`` survives and concurrent sweeps on one matrix do not disturb each other."""
    ranks = sorted({int(r) for r in bundle["ranks"]})
    bundle["classes"] = []
    if geometry_classes and len(ranks) > 1 and bundle.get("engine_factory") is None:
        bundle["classes"] = rank_classes(ranks, geometry_classes)
    return bundle["classes"]


def vb_factorize(mat, ranks=2, nrun=1, verbose=2, progress_bar=True, initializer="random", Itmax=10000,
                 hyper_update=(True, True, True, True), gamma_a=1, gamma_b=1, Tol=1e-5,
                 hyper_update_n0=10, hyper_update_dn=1, connectivity=False, fudge=None, ncores=1,
                 useC=True, unif_stop=True, seed=None, device=0, engine_factory=None, device_loop=True, concurrent=1,
                 device_init=False, geometry_classes=1, batch=None, grid=None, across_ranks=None, pad_rank=None):
    """Bayesian NMF of a count matrix on the MI355X engine; reference R/bayesian.R:229-301.

    ``mat`` is the genes x cells count matrix (dense, scipy sparse, or ``CountMatrix``).
    ``connectivity`` (reference default TRUE) is post-processing outside this path and is
    not computed; ``ncores`` / Rmpi is replaced by ``ccfindr_amd.parallel``; ``useC`` selects
    nothing (the native engine is the only backend); ``progress_bar`` is unused, as in the
    reference.  ``seed`` seeds the numpy Generator of the ``random`` initialiser (one stream
    per (run, rank)).  ``engine_factory(mat, rank)`` replaces the engine constructor (used by
    ``ccfindr_amd.parallel`` for cell-partitioned engines, and by the CPU tests of this loop).
    ``concurrent`` > 1 keeps that many (run, rank) units in flight on the one GPU, each on its own engine and HIP
    stream from its own host thread: on small matrices a step is latency bound (tens of microseconds with most of
    the chip idle), so independent factorisations overlap almost for free.  Every unit draws from its own seeded
    stream, so the result does not depend on ``concurrent``; with ``unif_stop`` a run's ranks beyond a constant
    basis column are still computed (and then discarded, as the sharded driver does).
    ``device_init`` draws the ``random`` initial state on the GPU (``vbnmf_engine_random_state``: Philox counters +
    Marsaglia-Tsang, one key per (seed, run, rank)) instead of with numpy on the host; off by default so that runs
    with an injected engine and runs on the HIP engine start from the same arrays.
    ``batch``: how many (run, rank) units are stepped by ONE launch (``engine.run_batch`` / ``vbnmf_batch_run``; 1: one loop at a
    time; None: ``batch_eligible`` -- up to 16 on small matrices): on the matrices the reference ships one loop cannot fill the
    GPU.  ``across_ranks`` (None: up to 2e6 stored entries): a batch also spans consecutive ranks of the rank loop, every engine
    made as wide as the widest rank's (``pad_rank``), so that a rank sweep with ``nrun`` = 1 batches as well.  ``grid``, ``pad_rank``:
    the launch grids and the row width of every engine (``VBEngine``); they fix the order of the block-wise sums, so runs on
    different grids or widths -- and hence ``batch=1`` against the default -- agree to rounding, not bit for bit.
    ``geometry_classes``: see ``plan_geometry`` (several ranks share the tiled layouts of the largest one; 0 = off).
    With shared layouts a rank's numbers depend, in the last bits, on WHICH ranks are in the sweep: the geometry fixes the
    order in which a step's partial sums are added (every result stays inside the 1e-12 / 1e-10 tolerances of one step,
    and any given sweep is bit-reproducible).  To re-run one rank of a sweep and get the same bits, pass the same
    ``ranks`` list again or use ``geometry_classes=0`` (every rank in its own geometry) for both runs.
    """
    del progress_bar, useC, ncores
    if connectivity:
        warnings.warn("connectivity/dispersion are outside the VB update path and are not computed")
    bundle = make_bundle(mat, ranks, nrun, verbose, initializer, Itmax, hyper_update, gamma_a, gamma_b, Tol,
                         hyper_update_n0, hyper_update_dn, fudge, unif_stop, seed, device, engine_factory)
    bundle["device_loop"] = bool(device_loop)      # False: step from the host (the loop below, literally)
    bundle["device_init"] = bool(device_init)
    bundle["concurrent"] = max(1, int(concurrent))
    bundle["grid"] = grid                               # (sweep workgroups, update blocks) of every engine; None: one per CU
    bundle["pad_rank"] = pad_rank                       # row width of every engine (a padded rank >= the widest rank's); None: each rank's own
    bundle["engines"] = {} if (nrun > 1 or bundle["concurrent"] > 1) else None   # restarts of a rank reuse its engine
    plan_geometry(bundle, geometry_classes)
    try:
        if bundle["concurrent"] > 1:
            from concurrent.futures import ThreadPoolExecutor
            if engine_factory is None and not getattr(bundle["mat"], "is_shell", False):
                # several units in flight: cut (and upload) the sweep's layouts once, here, instead of letting the first
                # units' threads cut the same pair side by side
                from .engine import sweep_workgroups
                n_wg = sweep_workgroups(device)
                # (the two sides of a geometry side by side, as engine creation cuts them)
                pieces = [(side, g) for g in sorted({geometry_rank_for(r, bundle["classes"]) or int(r) for r in bundle["ranks"]})
                          for side in (1, 0)]
                with ThreadPoolExecutor(max_workers=2) as cutters:
                    list(cutters.map(lambda pc: bundle["mat"].preload_layout(pc[0], pc[1], n_wg, device), pieces))
            units = [(irun, int(r)) for irun in range(1, nrun + 1) for r in bundle["ranks"]]
            units.sort(key=lambda u: -u[1])                                      # longest first
            with ThreadPoolExecutor(max_workers=bundle["concurrent"]) as pool:
                recs = list(pool.map(lambda u: vb_run_rank(u[0], u[1], bundle), units))
            records = dict(zip(units, recs))
            vb = [assemble_run({r: records[(irun, int(r))] for r in bundle["ranks"]}, [int(r) for r in bundle["ranks"]], unif_stop)
                  for irun in range(1, nrun + 1)]
        elif batch_eligible(bundle, batch, across_ranks) > 1:
            from .engine import batch_grid, padded_rank
            nb = batch_eligible(bundle, batch, across_ranks)
            bundle["grid"] = batch_grid(nb) if grid is None else grid          # B engines x 256 / B workgroups: one launch fills the chip
            if batch_across_ranks(bundle, across_ranks):                         # every rank's engine as wide as the widest rank's:
                bundle["pad_rank"] = padded_rank(max(bundle["ranks"]))          # one batch may then span ranks
            vb = vb_iterate_batched(bundle, nb)                                  # :260-261 (and :316): many loops, one launch
        else:
            vb = [vb_iterate(irun, bundle) for irun in range(1, nrun + 1)]       # :260-261
    finally:
        _close_engines(bundle)
    return select_best(vb, bundle["ranks"])
