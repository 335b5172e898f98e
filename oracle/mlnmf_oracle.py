"""CPU oracle for ccfindR's maximum-likelihood NMF path (factorize()) -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s CPU-baseline leg may import this module; the
product path (``ccfindr_amd``) never does.

PARITY UNPINNED: the reference holds no tests / golden vectors for this path and R is not in this image, so its
R code cannot be run.  Two restatements live here and are checked against each other
(``tests/test_oracle_mlnmf.py``):

* ``nmf_update_literal`` / ``likelihood_literal``: numpy, dense, statement by statement after
  ``R/factorize.R:2-27`` and ``:40-49``.
* ``update_csc``: ctypes front-end to ``mlnmf_oracle.c``, the stored-entries form (OpenMP) used at sizes the dense
  form cannot hold.

Driver-side restatements used to check the product's host loop: ``init`` (``R/factorize.R:30-38``),
``connectivity`` (``:51-60``), ``dispersion`` (``:62-67``), ``cophenet`` (``:69-78``), and ``factorize_run``, the
inner loop of ``factorize()`` for one run (``:190-217``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmlnmf_oracle.so")
_lib = None
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
EPS = float(np.finfo(np.float64).eps)          # .Machine$double.eps


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "mlnmf_oracle.c")
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
            subprocess.run(["make", "-C", _HERE, "-s"], check=True, stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_mlnmf_update_csc.restype = ctypes.c_int
        L.oracle_mlnmf_update_csc.argtypes = (
            [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, _ip, _ip, _dp, _dp, _dp, ctypes.c_int32,
             ctypes.c_double, ctypes.c_double, _dp, _dp, _dp, ctypes.c_int32])
        _lib = L
    return _lib


def nmf_update_literal(x, w, h, prior=False, gamma_a=1.0, gamma_b=1.0):
    """nmf_updateR(x, w, h, n, m, r, prior, gamma.a, gamma.b), R/factorize.R:2-27."""
    x = np.asarray(x, dtype=np.float64)                      # :4
    w = np.asarray(w, dtype=np.float64)                      # :5
    h = np.asarray(h, dtype=np.float64)                      # :6
    n, m = x.shape
    up = h * (w.T @ (x / (w @ h)))                           # :8
    down = np.repeat(w.sum(axis=0)[:, None], m, axis=1)      # :9   colSums(w), one per row k
    if prior:
        up = up + gamma_a - 1                                # :11
        down = down + gamma_a / gamma_b                      # :12
    h = up / down                                            # :14
    h[h < EPS] = EPS                                         # :15
    up = w * ((x / (w @ h)) @ h.T)                           # :17
    down = np.repeat(h.sum(axis=1)[None, :], n, axis=0)      # :18  rowSums(h), one per column k
    if prior:
        up = up + gamma_a - 1                                # :20
        down = down + gamma_a / gamma_b                      # :21
    w = up / down                                            # :23
    w[w < EPS] = EPS                                         # :24
    return {"ew": w, "eh": h}                                # :26


def likelihood_literal(mat, w, h):
    """likelihood(mat, w, h), R/factorize.R:40-49."""
    mat = np.asarray(mat, dtype=np.float64)
    wh = (np.asarray(w) @ np.asarray(h)).ravel(order="F")    # :42
    amat = mat.ravel(order="F")                              # :43
    with np.errstate(divide="ignore", invalid="ignore"):
        x = np.sum(amat * np.log(wh) - wh)                   # :44
    z = amat[amat > 0]                                       # :45
    x = x + np.sum(-z * np.log(z) + z)                       # :46
    return float(x / mat.shape[0] / mat.shape[1])            # :47


def update_csc(n, m, p, i, x, w, h, prior=False, gamma_a=1.0, gamma_b=1.0, nthreads=1):
    """One nmf_updateR step + likelihood on dgCMatrix slots (stored entries only; C, OpenMP)."""
    p = np.ascontiguousarray(p, dtype=np.int32)
    i = np.ascontiguousarray(i, dtype=np.int32)
    x = np.ascontiguousarray(x, dtype=np.float64)
    w0 = np.asfortranarray(w, dtype=np.float64)
    h0 = np.asfortranarray(h, dtype=np.float64)
    r = w0.shape[1]
    assert w0.shape == (n, r) and h0.shape == (r, m) and r <= 64
    w1 = np.empty((n, r), order="F")
    h1 = np.empty((r, m), order="F")
    lk = ctypes.c_double()
    rc = lib().oracle_mlnmf_update_csc(n, m, r, p.ctypes.data_as(_ip), i.ctypes.data_as(_ip), x.ctypes.data_as(_dp),
                                       w0.ctypes.data_as(_dp), h0.ctypes.data_as(_dp), int(bool(prior)),
                                       float(gamma_a), float(gamma_b), w1.ctypes.data_as(_dp), h1.ctypes.data_as(_dp),
                                       ctypes.byref(lk), int(nthreads))
    if rc != 0:
        raise RuntimeError("oracle_mlnmf_update_csc failed")
    return {"ew": w1, "eh": h1, "lk": lk.value}


def init(nrow, ncol, rank, rng):
    """init(), R/factorize.R:30-38: uniform(0,1) factors (numpy Generator in place of R's RNG stream)."""
    w = rng.uniform(size=(nrow, rank))
    h = rng.uniform(size=(rank, ncol))
    return {"ew": w, "eh": h}


def connectivity(h):
    """connectivity(h), R/factorize.R:51-60: for every pair j1 < j2 of cells, same arg-max row or not.
    Order of the returned vector = t(cnn)[lower.tri(t(cnn))], i.e. pairs (row, col) of the lower triangle in
    column-major order; cnn is symmetric, so this is the strict lower triangle column by column."""
    cid = np.argmax(np.asarray(h), axis=0)                   # which.max: first maximum
    cnn = cid[:, None] == cid[None, :]
    il = np.tril_indices(cnn.shape[0], -1)
    order = np.lexsort((il[0], il[1]))                       # column-major walk of the lower triangle
    return cnn.T[il[0][order], il[1][order]]


def dispersion(cnn, nc):
    """dispersion(cnn, nc), R/factorize.R:62-67."""
    con = np.sum((np.asarray(cnn, dtype=np.float64) - 0.5) ** 2)
    return float(1.0 / nc + 8.0 * con / nc ** 2)


def cophenet(conav, nc, method="average"):
    """cophenet(conav, nc, method), R/factorize.R:69-78 (scipy hclust / cophenetic in place of R's stats)."""
    from scipy.cluster.hierarchy import linkage, cophenet as sc_cophenet
    d = 1.0 - np.asarray(conav, dtype=np.float64)            # condensed distances; R fills tmp[lower.tri] column-major,
    Z = linkage(d, method=method)                            # which for a dist object is the same condensed order
    c, _ = sc_cophenet(Z, d)
    return float(c)


def factorize_run(update, mat, wh, Itmax=10000, Tol=1e-5, criterion="likelihood", ncnn_step=40):
    """Inner loop of factorize() for one run, R/factorize.R:190-217.  update(w, h) -> {ew, eh[, lk]}."""
    zstep = 0
    lkold = -np.inf
    cnn0 = None
    it = 0
    lk0 = np.nan
    for it in range(1, Itmax + 1):
        wh = update(wh["ew"], wh["eh"])                      # :195
        lk0 = wh["lk"] if "lk" in wh else likelihood_literal(mat, wh["ew"], wh["eh"])   # :196
        if criterion == "connectivity":
            cnn = connectivity(wh["eh"])
            nchange = cnn.size if it == 1 else int(np.sum(cnn != cnn0))
            zstep = zstep + 1 if nchange == 0 else 0
            if zstep == ncnn_step:
                break
            cnn0 = cnn
        elif criterion == "likelihood":
            if abs(lkold - lk0) < Tol * abs(lkold):          # :211 (lkold = -Inf on the first pass: Inf < Inf is FALSE)
                break
            lkold = lk0
        else:
            raise ValueError("Unknown stopping criterion.")
    return {"ew": wh["ew"], "eh": wh["eh"], "lk": lk0, "it": it}
