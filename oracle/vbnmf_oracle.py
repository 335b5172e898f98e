"""CPU oracle for the ccfindR VB-NMF update path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product path (``ccfindr_amd``) never does.

PARITY UNPINNED: the reference holds no tests / golden vectors for this path and cannot
be built or run in this image (needs R, Rcpp, RcppEigen, GSL).  Two independent
restatements live here and are checked against each other:

* ``update_dense`` / ``update_csc``: ctypes front-ends to ``vbnmf_oracle.c``, the literal
  restatement of ``src/vbnmf_update.cpp:19-101`` (dense) and its stored-entries form.
* ``update_rtwin``: numpy restatement of the R twin ``R/bayesian.R:56-106`` (reciprocal
  ``bew``/``beh``, scipy ``digamma``/``gammaln`` instead of the C file's own series).

Driver-side restatements (used to check the product's host loop):
``hyper_update`` (``R/bayesian.R:2-53``), ``vb_init_random`` (``R/bayesian.R:111-115,162-170``),
``vb_iterate`` (``R/bayesian.R:316-385``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libvbnmf_oracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    """Compile vbnmf_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "vbnmf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        for name in ("oracle_digamma", "oracle_trigamma", "oracle_lgamma"):
            f = getattr(L, name)
            f.restype = ctypes.c_double
            f.argtypes = [ctypes.c_double]
        L.oracle_max_threads.restype = ctypes.c_int
        L.oracle_vbnmf_update_dense.restype = ctypes.c_int
        L.oracle_vbnmf_update_dense.argtypes = (
            [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, _dp, _dp, _dp, _dp]
            + [ctypes.c_double] * 5 + [_dp] * 7)
        L.oracle_vbnmf_update_csc.restype = ctypes.c_int
        L.oracle_vbnmf_update_csc.argtypes = (
            [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, _ip, _ip, _dp, _dp, _dp, _dp]
            + [ctypes.c_double] * 5 + [_dp] * 7 + [ctypes.c_int32])
        _lib = L
    return _lib


def _f(a):
    """Column-major float64 copy (R / Eigen storage)."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _ptr(a):
    return a.ctypes.data_as(_dp)


def digamma(x):
    L = lib()
    return np.vectorize(L.oracle_digamma, otypes=[np.float64])(x)


def trigamma(x):
    L = lib()
    return np.vectorize(L.oracle_trigamma, otypes=[np.float64])(x)


def _outputs(n, m, r):
    lw = np.empty((n, r), order="F"); ew = np.empty((n, r), order="F"); dw = np.empty((n, r), order="F")
    lh = np.empty((r, m), order="F"); eh = np.empty((r, m), order="F"); dh = np.empty((r, m), order="F")
    return lw, lh, ew, eh, dw, dh


def _as_list(lw, lh, ew, eh, dw, dh, lkh):
    # key order of src/vbnmf_update.cpp:92-100
    return {"w": ew, "h": eh, "lw": lw, "lh": lh, "ew": ew, "eh": eh, "lkh": float(lkh), "dw": dw, "dh": dh}


def update_dense(X, wh, hyper, fudge=np.finfo(np.float64).eps):
    """vbnmf_update(X, wh, hyper, fudge) -- dense literal restatement (C)."""
    X = _f(X)
    n, m = X.shape
    lw0, lh0, eh0 = _f(wh["lw"]), _f(wh["lh"]), _f(wh["eh"])
    r = lw0.shape[1]
    assert lw0.shape == (n, r) and lh0.shape == (r, m) and eh0.shape == (r, m)
    lw, lh, ew, eh, dw, dh = _outputs(n, m, r)
    lkh = ctypes.c_double()
    rc = lib().oracle_vbnmf_update_dense(
        n, m, r, _ptr(X), _ptr(lw0), _ptr(lh0), _ptr(eh0),
        hyper["aw"], hyper["bw"], hyper["ah"], hyper["bh"], float(fudge),
        _ptr(lw), _ptr(lh), _ptr(ew), _ptr(eh), _ptr(dw), _ptr(dh), ctypes.byref(lkh))
    if rc != 0:
        raise RuntimeError("oracle_vbnmf_update_dense failed")
    return _as_list(lw, lh, ew, eh, dw, dh, lkh.value)


def update_csc(n, m, p, i, x, wh, hyper, fudge=np.finfo(np.float64).eps, nthreads=1):
    """Same step with X as dgCMatrix slots (p, i, x); stored entries only (C, OpenMP)."""
    p = np.ascontiguousarray(p, dtype=np.int32)
    i = np.ascontiguousarray(i, dtype=np.int32)
    x = np.ascontiguousarray(x, dtype=np.float64)
    lw0, lh0, eh0 = _f(wh["lw"]), _f(wh["lh"]), _f(wh["eh"])
    r = lw0.shape[1]
    assert lw0.shape == (n, r) and lh0.shape == (r, m) and eh0.shape == (r, m)
    lw, lh, ew, eh, dw, dh = _outputs(n, m, r)
    lkh = ctypes.c_double()
    rc = lib().oracle_vbnmf_update_csc(
        n, m, r, p.ctypes.data_as(_ip), i.ctypes.data_as(_ip), _ptr(x),
        _ptr(lw0), _ptr(lh0), _ptr(eh0),
        hyper["aw"], hyper["bw"], hyper["ah"], hyper["bh"], float(fudge),
        _ptr(lw), _ptr(lh), _ptr(ew), _ptr(eh), _ptr(dw), _ptr(dh), ctypes.byref(lkh), int(nthreads))
    if rc != 0:
        raise RuntimeError("oracle_vbnmf_update_csc failed")
    return _as_list(lw, lh, ew, eh, dw, dh, lkh.value)


def update_rtwin(x, wh, hyper, fudge=None):
    """numpy restatement of vbnmf_updateR, R/bayesian.R:56-106 (independent of the C file)."""
    from scipy.special import digamma as sp_digamma, gammaln
    x = np.asarray(x, dtype=np.float64)
    n, m = x.shape
    lw = np.asarray(wh["lw"], dtype=np.float64)
    lh = np.asarray(wh["lh"], dtype=np.float64)
    eh = np.asarray(wh["eh"], dtype=np.float64)
    aw, bw, ah, bh = hyper["aw"], hyper["bw"], hyper["ah"], hyper["bh"]
    wth = lw @ lh                                              # :71
    sw = lw * ((x / wth) @ lh.T)                               # :72
    sh = lh * (lw.T @ (x / wth))                               # :73
    alw = aw + sw                                              # :75
    bew = 1.0 / (aw / bw + np.tile(eh.sum(axis=1), (n, 1)))    # :76
    ew = alw * bew                                             # :77
    alh = ah + sh                                              # :79
    beh = 1.0 / (ah / bh + np.tile(ew.sum(axis=0)[:, None], (1, m)))  # :80
    eh = alh * beh                                             # :81
    lw = np.exp(sp_digamma(alw)) * bew                         # :83
    lh = np.exp(sp_digamma(alh)) * beh                         # :84
    if fudge is None:
        fudge = np.finfo(np.float64).eps                       # :85
    lw[lw < fudge] = fudge                                     # :86
    lh[lh < fudge] = fudge                                     # :87
    wth = lw @ lh                                              # :89
    U1 = -ew @ eh - gammaln(x + 1) - x * ((((lw * np.log(lw)) @ lh) + lw @ (lh * np.log(lh))) / wth
                                          - np.log(wth))       # :90-91
    U2 = -(aw / bw) * ew - gammaln(aw) + aw * np.log(aw / bw) + alw * (1 + np.log(bew)) + gammaln(alw)
    U3 = -(ah / bh) * eh - gammaln(ah) + ah * np.log(ah / bh) + alh * (1 + np.log(beh)) + gammaln(alh)
    U = (U1.sum() + U2.sum() + U3.sum()) / (float(n) * float(m))   # :96-97
    dw = alw * bew ** 2                                        # :102
    dh = alh * beh ** 2                                        # :103
    return _as_list(lw, lh, ew, eh, dw, dh, U)


def hyper_update(flags, wh, hyper, Niter=100, Tol=1e-4):
    """R/bayesian.R:2-53, statement by statement (scipy digamma / polygamma)."""
    from scipy.special import digamma as sp_digamma, polygamma
    flags = [bool(f) for f in flags]
    if sum(flags) == 0:
        return dict(hyper)
    aw0, ah0 = hyper["aw"], hyper["ah"]
    lwm = float(np.mean(np.log(wh["lw"])))
    lhm = float(np.mean(np.log(wh["lh"])))
    ewm = float(np.mean(wh["ew"]))
    ehm = float(np.mean(wh["eh"]))
    bw0, bh0 = hyper["bw"], hyper["bh"]
    if flags[0] + flags[2] > 0:
        i = 1
        while i < Niter:
            dw = ((np.log(aw0) - sp_digamma(aw0) - ewm / bw0 + 1 + lwm - np.log(bw0))
                  / (1 / aw0 - polygamma(1, aw0))) if flags[0] else 0.0
            dh = ((np.log(ah0) - sp_digamma(ah0) - ehm / bh0 + 1 + lhm - np.log(bh0))
                  / (1 / ah0 - polygamma(1, ah0))) if flags[2] else 0.0
            aw1, ah1 = aw0 - dw, ah0 - dh
            while aw1 <= 0:
                dw /= 2
                aw1 = aw0 - dw
            while ah1 <= 0:
                dh /= 2
                ah1 = ah0 - dh
            df = (1 - aw1 / aw0) ** 2 + (1 - ah1 / ah0) ** 2
            if df < Tol:
                break
            aw0, ah0 = aw1, ah1
            i += 1
        if i == Niter:
            raise RuntimeError("Hyper-parameter update failed to converge")
    else:
        aw1, ah1 = aw0, ah0
    bw1 = ewm if flags[1] else bw0
    bh1 = ehm          # :50-51: both branches assign ehm
    return {"aw": float(aw1), "bw": float(bw1), "ah": float(ah1), "bh": float(bh1)}


def vb_init_random(n, m, r, hyper, rng):
    """'random' initialiser, R/bayesian.R:111-115,162-170, with a numpy Generator standing in
    for R's RNG stream (which cannot be reproduced without R)."""
    w = rng.gamma(shape=hyper["aw"], scale=hyper["bw"] / hyper["aw"], size=(n, r))
    h = rng.gamma(shape=hyper["ah"], scale=hyper["bh"] / hyper["ah"], size=(r, m))
    return {"w": w, "h": h, "lw": w.copy(), "lh": h.copy(), "ew": w.copy(), "eh": h.copy(),
            "dw": np.zeros((n, r)), "dh": np.zeros((r, m))}


def vb_iterate(update, wh, hyper, Itmax=10000, Tol=1e-5, hyper_flags=(True,) * 4,
               n0=10, dn=1, fudge=np.finfo(np.float64).eps):
    """The per-rank loop of R/bayesian.R:336-352.  ``update(wh, hyper, fudge) -> list``.

    Returns (wh, hyper, lk0, it, trace) where lk0 lags on a convergence break exactly
    as :346-348 make it (the break precedes ``lk0 <- wh$lkh``).
    """
    lk0 = 0.0
    trace = []
    it = 0
    for it in range(1, Itmax + 1):
        wh = update(wh, hyper, fudge)                                       # :339
        if it > n0 and it % dn == 0:                                        # :342
            hyper = hyper_update(hyper_flags, wh, hyper, Niter=100, Tol=1e-3)
        trace.append((wh["lkh"], dict(hyper)))
        if np.isnan(wh["lkh"]):                                             # :345
            break
        if it > 1 and it > n0 and wh["lkh"] >= lk0 and abs(1 - wh["lkh"] / lk0) < Tol:  # :346-347
            break
        lk0 = wh["lkh"]                                                     # :348
    return wh, hyper, lk0, it, trace


def vb_init_svd(mat, rank):
    """vb_init(..., initializer='svd'), R/bayesian.R:116-149, element by element as the R code walks it
    (vapply per entry), quirks included (:135-136 take the norms of xp, yp again)."""
    A = np.asarray(mat.toarray() if hasattr(mat, "toarray") else mat, dtype=np.float64)
    nrow, ncol = A.shape
    U, D, Vt = np.linalg.svd(A, full_matrices=False)
    w = np.zeros((nrow, rank)); h = np.zeros((rank, ncol))
    d1 = D[0] ** 0.5
    for i in range(nrow):
        w[i, 0] = d1 * U[i, 0]
    sgn = int(w[0, 0] > 0) - int(w[0, 0] < 0)
    if sgn < 0:
        w = -w
    for j in range(ncol):
        h[0, j] = sgn * d1 * Vt[0, j]
    for k in range(1, rank):
        x, y = U[:, k], Vt[k, :]
        xp = np.array([v if v > 0 else 0.0 for v in x]); yp = np.array([v if v > 0 else 0.0 for v in y])
        xn = np.array([-v if v < 0 else 0.0 for v in x]); yn = np.array([-v if v < 0 else 0.0 for v in y])
        xpnrm = sum(v * v for v in xp) ** 0.5; ypnrm = sum(v * v for v in yp) ** 0.5
        mp = xpnrm * ypnrm
        xnnrm = sum(v * v for v in xp) ** 0.5; ynnrm = sum(v * v for v in yp) ** 0.5
        mn = xnnrm * ynnrm
        if mp >= mn:
            u, v, sig = xp / xpnrm, yp / ypnrm, mp
        else:
            u, v, sig = xn / xnnrm, yn / ynnrm, mn
        w[:, k] = (D[k] * sig) ** 0.5 * u
        h[k, :] = (D[k] * sig) ** 0.5 * v
    return {"w": w, "h": h, "lw": w.copy(), "lh": h.copy(), "ew": w.copy(), "eh": h.copy(),
            "dw": np.zeros((nrow, rank)), "dh": np.zeros((rank, ncol))}
