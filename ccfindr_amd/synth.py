"""Synthetic count matrices shaped like the reference's simulators (numpy, deterministic).

* ``simulate_data``  Dirichlet-multinomial cluster mixture; reference R/utils.R:757-797
  (the non-``generate.factors`` branch, :787-795): per cluster a gene distribution
  q_k ~ Dirichlet(alpha0 * 1_n), each cell ~ Multinomial(L_j, q_k), columns shuffled.
* ``simulate_whx``   Gamma-Poisson from the priors; reference R/utils.R:826-846.

numpy's Generator stands in for R's RNG; these produce data of the same law, not the same draws.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def simulate_data(nfeatures, nsamples, nfactor=10, alpha0=0.5, shuffle=True, seed=0, depth=None, sparse=True):
    """genes x cells counts.  ``nsamples``: cells per cluster.  ``depth``: per-cell total count
    (array of sum(nsamples)) -- default nfeatures*nfactor as the reference (R/utils.R:790)."""
    rng = np.random.default_rng(seed)
    nsamples = [int(s) for s in nsamples]
    m = int(sum(nsamples))
    n = int(nfeatures)
    if depth is None:
        depth = np.full(m, n * nfactor, dtype=np.int64)
    depth = np.asarray(depth, dtype=np.int64)
    keys = []
    c0 = 0
    perm = rng.permutation(m).astype(np.int64) if shuffle else np.arange(m, dtype=np.int64)
    for k, mk in enumerate(nsamples):
        q = rng.dirichlet(np.full(n, alpha0))
        Lk = depth[c0:c0 + mk]
        # Multinomial(L_j, q) for every cell of the cluster at once: the gene of each of the
        # sum(L) molecules is iid ~ q, so draw the cluster's per-gene totals in one multinomial
        # and deal the molecules to cells by a uniform shuffle (same law, O(n + sum L)).
        tot = rng.multinomial(int(Lk.sum()), q)
        g = np.repeat(np.arange(n, dtype=np.int64), tot)
        rng.shuffle(g)
        c = np.repeat(perm[c0:c0 + mk], Lk)
        keys.append(c * n + g)                   # column-major linear index of each molecule
        c0 += mk
    keys = np.concatenate(keys)
    keys.sort()
    first = np.flatnonzero(np.concatenate(([True], keys[1:] != keys[:-1])))
    cnt = np.diff(np.concatenate((first, [keys.size]))).astype(np.float64)
    uk = keys[first]
    col = uk // n
    row = (uk - col * n).astype(np.int32)
    indptr = np.cumsum(np.bincount(col + 1, minlength=m + 1))
    X = sp.csc_matrix((cnt, row, indptr.astype(np.int32)), shape=(n, m))
    return X if sparse else np.asfortranarray(X.toarray())


def fill_empty(X, seed=0):
    """Give every all-zero row / column one count at a random position, keeping the shape
    (the other way to satisfy the reference's guards, R/bayesian.R:244-247)."""
    rng = np.random.default_rng(seed)
    X = sp.csc_matrix(X) if not sp.issparse(X) else X.tocsc()
    n, m = X.shape
    er = np.flatnonzero(np.asarray(X.sum(axis=1)).ravel() == 0)
    ec = np.flatnonzero(np.asarray(X.sum(axis=0)).ravel() == 0)
    if er.size == 0 and ec.size == 0:
        return X
    rows = np.concatenate((er, rng.integers(0, n, ec.size)))
    cols = np.concatenate((rng.integers(0, m, er.size), ec))
    X = (X + sp.csc_matrix((np.ones(rows.size), (rows, cols)), shape=(n, m))).tocsc()
    X.sort_indices()
    return X


def drop_empty(X):
    """Remove all-zero rows/columns (vb_factorize refuses them, reference R/bayesian.R:244-247)."""
    if sp.issparse(X):
        X = X.tocsc()
        keep_r = np.asarray(X.sum(axis=1)).ravel() > 0
        keep_c = np.asarray(X.sum(axis=0)).ravel() > 0
        return X[keep_r][:, keep_c].tocsc()
    keep_r = X.sum(axis=1) > 0
    keep_c = X.sum(axis=0) > 0
    return np.asfortranarray(X[keep_r][:, keep_c])


def simulate_whx(nrow, ncol, rank, aw=0.1, bw=1.0, ah=0.1, bh=1.0, seed=0):
    rng = np.random.default_rng(seed)
    w = rng.gamma(shape=aw, scale=bw / aw, size=(nrow, rank))
    h = rng.gamma(shape=ah, scale=bh / ah, size=(rank, ncol))
    x = rng.poisson(w @ h).astype(np.float64)
    i = x.sum(axis=1) > 0
    j = x.sum(axis=0) > 0
    return {"w": w[i], "h": h[:, j], "x": np.asfortranarray(x[i][:, j])}


def random_state(n, m, r, hyper=None, seed=0):
    """Initial (lw, lh, eh) drawn like vb_init('random') (reference R/bayesian.R:111-115)."""
    hyper = hyper or {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    rng = np.random.default_rng(seed)
    w = rng.gamma(shape=hyper["aw"], scale=hyper["bw"] / hyper["aw"], size=(n, r))
    h = rng.gamma(shape=hyper["ah"], scale=hyper["bh"] / hyper["ah"], size=(r, m))
    return {"lw": w, "lh": h, "ew": w.copy(), "eh": h.copy()}
