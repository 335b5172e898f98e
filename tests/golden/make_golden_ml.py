#!/usr/bin/env python3
"""Generates the ML-NMF golden fixtures tests/golden/ml_*.npz from the CPU oracle.

PARITY UNPINNED: the reference holds no tests or stored outputs for factorize()'s step (R/factorize.R:2-27,
:40-49) and R is not in this image, so these vectors pin the ORACLE, not the reference: every case is produced by
the dense literal restatement (oracle/mlnmf_oracle.py) and only written after the stored-entries C form
(oracle/mlnmf_oracle.c) agrees with it to 1e-13.  Inputs come from numpy's PCG64 with the seeds below; the PBMC
case reuses the count data of tests/golden/pbmc_extdata_r5.npz (the reference's bundled sample, data only).

    python tests/golden/make_golden_ml.py          # rewrites tests/golden/ml_*.npz
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import mlnmf_oracle as O              # noqa: E402


def counts(n, m, lam, seed):
    rng = np.random.default_rng(seed)
    X = rng.poisson(lam, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1
    X[rng.integers(0, n, m), np.arange(m)] += 1
    return np.asfortranarray(X)


def agree(X, w, h, prior, ga, gb):
    a = O.nmf_update_literal(X, w, h, prior, ga, gb)
    lk = O.likelihood_literal(X, a["ew"], a["eh"])
    S = sp.csc_matrix(X)
    b = O.update_csc(X.shape[0], X.shape[1], S.indptr, S.indices, S.data, w, h, prior, ga, gb, nthreads=2)
    for k in ("ew", "eh"):
        err = np.max(np.abs(a[k] - b[k]) / np.abs(a[k]))
        assert err < 1e-13, (k, err)
    assert abs(lk / b["lk"] - 1) < 1e-13, (lk, b["lk"])
    return a, lk


def main():
    cases = {
        "dense_64x96_r4": (counts(64, 96, 1.5, 11), 4, False, 1.0, 1.0, 12),
        "sparse5pct_300x400_r10": (counts(300, 400, 0.045, 13), 10, False, 1.0, 1.0, 14),
        "prior_90x140_r5": (counts(90, 140, 0.6, 21), 5, True, 2.5, 0.7, 22),
        "rank1_50x70_r1": (counts(50, 70, 1.0, 23), 1, False, 1.0, 1.0, 24),
    }
    Xn = counts(80, 150, 0.7, 15)
    cases["noninteger_80x150_r5"] = (Xn * (np.median(Xn.sum(axis=0)) / Xn.sum(axis=0))[None, :], 5, False, 1.0, 1.0, 16)
    for name, (X, r, prior, ga, gb, seed) in cases.items():
        n, m = X.shape
        wh = O.init(n, m, r, np.random.default_rng(seed))
        a, lk = agree(X, wh["ew"], wh["eh"], prior, ga, gb)
        np.savez_compressed(os.path.join(HERE, f"ml_step_{name}.npz"), X=X, r=r, prior=prior, gamma=np.array([ga, gb]),
                            w0=wh["ew"], h0=wh["eh"], ew=a["ew"], eh=a["eh"], lk=lk)
        print(f"ml_step_{name}: n={n} m={m} r={r} lk={lk:.15g}")

    # a run of factorize()'s inner loop (likelihood criterion, :190-217): trajectory, stop iteration, final pair
    X = counts(120, 200, 0.5, 31)
    wh = O.init(120, 200, 3, np.random.default_rng(32))
    w, h, lks = wh["ew"], wh["eh"], []
    for _ in range(60):
        o = O.nmf_update_literal(X, w, h)
        w, h = o["ew"], o["eh"]
        lks.append(O.likelihood_literal(X, w, h))
    run = O.factorize_run(lambda a, b: O.nmf_update_literal(X, a, b), X, wh, Itmax=2000, Tol=1e-6)
    np.savez_compressed(os.path.join(HERE, "ml_traj_120x200_r3.npz"), X=X, r=3, w0=wh["ew"], h0=wh["eh"], lk=np.array(lks),
                        ew60=w, eh60=h, it=run["it"], lk_stop=run["lk"], ew_stop=run["ew"], eh_stop=run["eh"], tol=1e-6)
    print(f"ml_traj: lk[0]={lks[0]:.12g} lk[59]={lks[-1]:.12g} stop it={run['it']} lk={run['lk']:.12g}")

    # the reference's bundled PBMC sample (data only, from the existing fixture)
    p = os.path.join(HERE, "pbmc_extdata_r5.npz")
    if os.path.exists(p):
        z = np.load(p)
        n, m = int(z["n"]), int(z["m"])
        S = sp.csc_matrix((z["data"].astype(np.float64), z["indices"], z["indptr"]), shape=(n, m))
        wh = O.init(n, m, 5, np.random.default_rng(41))
        w, h, lks = wh["ew"], wh["eh"], []
        for _ in range(20):
            o = O.update_csc(n, m, S.indptr, S.indices, S.data, w, h)
            w, h = o["ew"], o["eh"]
            lks.append(o["lk"])
        d = O.nmf_update_literal(S.toarray(), wh["ew"], wh["eh"])
        # 4.6e5 dense terms against the collapsed stored-entries sum: summation order shows at ~3e-13
        assert abs(O.likelihood_literal(S.toarray(), d["ew"], d["eh"]) / lks[0] - 1) < 1e-11
        np.savez_compressed(os.path.join(HERE, "ml_pbmc_extdata_r5.npz"), r=5, w0=wh["ew"], h0=wh["eh"], lk=np.array(lks),
                            ew20=w, eh20=h)
        print(f"ml_pbmc: lk[0]={lks[0]:.12g} lk[19]={lks[-1]:.12g}")


if __name__ == "__main__":
    main()
