#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

PARITY UNPINNED: the reference (ccfindR v1.5.1) holds no tests or stored outputs for
vbnmf_update and cannot be built or run here (needs R, Rcpp, RcppEigen, GSL), so these
vectors pin the ORACLE, not the reference: every case is produced by the literal C
restatement of src/vbnmf_update.cpp (oracle/vbnmf_oracle.c) and is only written after the
independent numpy restatement of the R twin (R/bayesian.R:56-106) agrees with it to 1e-12
(factors) / 1e-12 (lkh).  Inputs come from numpy's PCG64 with the seeds below.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from ccfindr_amd import synth                     # noqa: E402  (generators only; no device code runs)
from oracle import vbnmf_oracle as O              # noqa: E402

EPS = float(np.finfo(np.float64).eps)
FACT = ("lw", "lh", "ew", "eh", "dw", "dh")


def counts(n, m, lam, seed):
    rng = np.random.default_rng(seed)
    X = rng.poisson(lam, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1
    X[rng.integers(0, n, m), np.arange(m)] += 1
    return np.asfortranarray(X)


def single_step_cases():
    hy1 = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    cases = {}
    X = synth.drop_empty(synth.simulate_data(200, (100, 150, 250), seed=1, sparse=False))
    cases["c1_200x500_r3"] = (X, 3, hy1, EPS, 1001)
    cases["dense_64x96_r4"] = (counts(64, 96, 1.5, 11), 4, hy1, EPS, 12)
    cases["sparse5pct_300x400_r10"] = (counts(300, 400, 0.045, 13), 10, hy1, EPS, 14)
    Xn = counts(80, 150, 0.7, 15)
    Xn = Xn * (np.median(Xn.sum(axis=0)) / Xn.sum(axis=0))[None, :]       # normalize_count style (R/utils.R:318-327)
    cases["noninteger_80x150_r5"] = (Xn, 5, hy1, EPS, 16)
    cases["smallshape_90x120_r4"] = (counts(90, 120, 0.3, 17), 4, {"aw": 0.05, "bw": 1.0, "ah": 0.05, "bh": 1.0}, EPS, 18)
    cases["fudge0_90x120_r4"] = (counts(90, 120, 0.3, 19), 4, hy1, 0.0, 20)
    cases["rank1_50x70_r1"] = (counts(50, 70, 1.0, 21), 1, {"aw": 2.5, "bw": 0.7, "ah": 0.3, "bh": 3.0}, EPS, 22)
    return cases


def main():
    for name, (X, r, hyper, fudge, seed) in single_step_cases().items():
        n, m = X.shape
        wh = synth.random_state(n, m, r, hyper, seed=seed)
        a = O.update_dense(X, wh, hyper, fudge)
        b = O.update_rtwin(X, wh, hyper, fudge)
        for k in FACT:
            err = np.max(np.abs(a[k] - b[k]) / np.abs(b[k]))
            assert err < 1e-12, (name, k, err)
        assert abs(a["lkh"] / b["lkh"] - 1) < 1e-12, (name, a["lkh"], b["lkh"])
        np.savez_compressed(os.path.join(HERE, f"step_{name}.npz"), X=X, r=r, fudge=fudge,
                            hyper=np.array([hyper[k] for k in ("aw", "bw", "ah", "bh")]),
                            lw0=wh["lw"], lh0=wh["lh"], eh0=wh["eh"], lkh=a["lkh"], **{k: a[k] for k in FACT})
        print(f"step_{name}: n={n} m={m} r={r} lkh={a['lkh']:.15g}")

    # trajectories on the C1-shaped matrix: fixed hyper, and the reference's default hyper updates
    X = synth.drop_empty(synth.simulate_data(200, (100, 150, 250), seed=1, sparse=False))
    n, m = X.shape
    hy1 = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    wh0 = synth.random_state(n, m, 3, hy1, seed=1001)
    upd = lambda wh, hy, fud: O.update_dense(X, wh, hy, fud)
    for tag, flags in (("fixed", (False,) * 4), ("hyper", (True,) * 4)):
        wh, hy, lk0, it, trace = O.vb_iterate(upd, dict(wh0), dict(hy1), Itmax=50, Tol=0.0, hyper_flags=flags)
        assert it == 50
        np.savez_compressed(os.path.join(HERE, f"traj_{tag}_c1_r3.npz"), X=X, r=3, lw0=wh0["lw"], lh0=wh0["lh"], eh0=wh0["eh"],
                            lkh=np.array([t[0] for t in trace]),
                            hyper=np.array([[t[1][k] for k in ("aw", "bw", "ah", "bh")] for t in trace]),
                            ew=wh["ew"], eh=wh["eh"], lk0=lk0, it=it)
        print(f"traj_{tag}: lkh[0]={trace[0][0]:.12g} lkh[-1]={trace[-1][0]:.12g} hyper[-1]={trace[-1][1]}")

    # the loop's stopping rule (R/bayesian.R:345-348): iteration count and the lagging lk0
    wh, hy, lk0, it, trace = O.vb_iterate(upd, dict(wh0), dict(hy1), Itmax=400, Tol=1e-5)
    np.savez_compressed(os.path.join(HERE, "loop_c1_r3.npz"), X=X, r=3, lw0=wh0["lw"], lh0=wh0["lh"], eh0=wh0["eh"],
                        it=it, lk0=lk0, lkh_last=trace[-1][0], hyper=np.array([hy[k] for k in ("aw", "bw", "ah", "bh")]),
                        ew=wh["ew"], eh=wh["eh"])
    print(f"loop: it={it} lk0={lk0:.12g} last lkh={trace[-1][0]:.12g}")


def pbmc_fixture():
    """The reference's bundled PBMC sample (inst/extdata/matrix.mtx: 1030 genes x 450 cells, 91 200
    stored counts) as a realistic input: DATA only (coordinates and counts), with one oracle step at
    rank 5 as the expected output.  Needs the reference checkout; skipped where it is absent."""
    path = "/root/reference/inst/extdata/matrix.mtx"
    if not os.path.exists(path):
        print("pbmc: reference data file not present, fixture left as is")
        return
    import scipy.sparse as sp
    rows, cols, vals = [], [], []
    with open(path) as f:
        header = None
        for line in f:
            if line.startswith("%"):
                continue
            a = line.split()
            if header is None:
                header = (int(a[0]), int(a[1]), int(a[2]))
                continue
            rows.append(int(a[0]) - 1); cols.append(int(a[1]) - 1); vals.append(float(a[2]))
    n, m, nnz = header
    assert len(vals) == nnz
    X = sp.csc_matrix((vals, (rows, cols)), shape=(n, m))
    keep_r = np.asarray(X.sum(axis=1)).ravel() > 0            # vb_factorize refuses empty rows (R/bayesian.R:244-247)
    X = X[keep_r].tocsc()
    X.sort_indices()
    n = X.shape[0]
    r = 5
    hy1 = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    wh = synth.random_state(n, m, r, hy1, seed=31)
    a = O.update_dense(X.toarray(), wh, hy1, EPS)
    b = O.update_csc(n, m, X.indptr, X.indices, X.data, wh, hy1, EPS)
    assert abs(a["lkh"] / b["lkh"] - 1) < 1e-12
    np.savez_compressed(os.path.join(HERE, "pbmc_extdata_r5.npz"), n=n, m=m, indptr=X.indptr.astype(np.int32),
                        indices=X.indices.astype(np.int32), data=X.data.astype(np.int32), r=r,
                        lw0=wh["lw"], lh0=wh["lh"], eh0=wh["eh"], lkh=a["lkh"], **{k: a[k] for k in FACT})
    print(f"pbmc: n={n} m={m} nnz={X.nnz} lkh={a['lkh']:.15g}")


if __name__ == "__main__":
    main()
    pbmc_fixture()
