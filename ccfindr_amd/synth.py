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
    rows, cols = [], []
    c0 = 0
    for k, mk in enumerate(nsamples):
        q = rng.dirichlet(np.full(n, alpha0))
        cdf = np.cumsum(q)
        cdf[-1] = 1.0
        Lk = depth[c0:c0 + mk]
        # multinomial draws by inverse-cdf on sum(L) uniforms: gene ids of every counted molecule
        u = rng.random(int(Lk.sum()))
        g = np.searchsorted(cdf, u, side="right").astype(np.int32)
        np.minimum(g, n - 1, out=g)
        c = np.repeat(np.arange(c0, c0 + mk, dtype=np.int32), Lk)
        rows.append(g)
        cols.append(c)
        c0 += mk
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    if shuffle:
        perm = rng.permutation(m).astype(np.int32)
        cols = perm[cols]
    X = sp.coo_matrix((np.ones(rows.size, dtype=np.float64), (rows, cols)), shape=(n, m)).tocsc()
    X.sum_duplicates()
    X.sort_indices()
    return X if sparse else np.asfortranarray(X.toarray())


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
