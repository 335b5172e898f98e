"""CPU: the two ML-NMF restatements (dense literal numpy / stored-entries C) against each other and against the
committed golden vectors (tests/golden/ml_*.npz), plus the driver-side helpers of reference R/factorize.R."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import mlnmf_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "ml_step_*.npz"))), ids=lambda p: os.path.basename(p)[8:-4])
def test_step_golden_both_forms(path):
    z = np.load(path)
    X, prior, (ga, gb) = z["X"], bool(z["prior"]), z["gamma"]
    a = O.nmf_update_literal(X, z["w0"], z["h0"], prior, ga, gb)
    assert np.array_equal(a["ew"], z["ew"]) and np.array_equal(a["eh"], z["eh"])
    assert O.likelihood_literal(X, a["ew"], a["eh"]) == float(z["lk"])
    S = sp.csc_matrix(X)
    for nt in (1, 3):
        b = O.update_csc(X.shape[0], X.shape[1], S.indptr, S.indices, S.data, z["w0"], z["h0"], prior, ga, gb, nthreads=nt)
        assert relerr(b["ew"], z["ew"]) < 1e-13 and relerr(b["eh"], z["eh"]) < 1e-13
        assert abs(b["lk"] / float(z["lk"]) - 1) < 1e-13


def test_trajectory_golden_and_stop_rule():
    z = np.load(os.path.join(GOLD, "ml_traj_120x200_r3.npz"))
    X = z["X"]
    S = sp.csc_matrix(X)
    w, h = z["w0"], z["h0"]
    for t in range(60):
        o = O.update_csc(120, 200, S.indptr, S.indices, S.data, w, h)
        w, h = o["ew"], o["eh"]
        assert abs(o["lk"] / z["lk"][t] - 1) < 1e-11
    assert relerr(w, z["ew60"]) < 1e-10 and relerr(h, z["eh60"]) < 1e-10
    assert np.all(np.diff(z["lk"]) >= 0)                      # multiplicative updates never lower the likelihood
    run = O.factorize_run(lambda a, b: O.update_csc(120, 200, S.indptr, S.indices, S.data, a, b), X,
                          {"ew": z["w0"], "eh": z["h0"]}, Itmax=2000, Tol=float(z["tol"]))
    assert run["it"] == int(z["it"]) and abs(run["lk"] / float(z["lk_stop"]) - 1) < 1e-10


def test_pbmc_golden():
    z = np.load(os.path.join(GOLD, "ml_pbmc_extdata_r5.npz"))
    d = np.load(os.path.join(GOLD, "pbmc_extdata_r5.npz"))
    n, m = int(d["n"]), int(d["m"])
    w, h = z["w0"], z["h0"]
    for t in range(20):
        o = O.update_csc(n, m, d["indptr"], d["indices"], d["data"].astype(np.float64), w, h, nthreads=2)
        w, h = o["ew"], o["eh"]
        assert abs(o["lk"] / z["lk"][t] - 1) < 1e-11
    assert relerr(w, z["ew20"]) < 1e-10 and relerr(h, z["eh20"]) < 1e-10


def test_likelihood_is_minus_kl_per_element():
    """likelihood = -(generalised KL divergence of x from wh) / (n m) (R/factorize.R:119): zero at x == wh."""
    rng = np.random.default_rng(0)
    w, h = rng.uniform(0.5, 2, (7, 2)), rng.uniform(0.5, 2, (2, 9))
    x = w @ h
    assert abs(O.likelihood_literal(x, w, h)) < 1e-14
    assert O.likelihood_literal(x, w * 1.3, h) < 0


def test_connectivity_order_and_measures():
    h = np.array([[3.0, 0.1, 2.0, 0.0], [1.0, 4.0, 2.0, 5.0]])     # arg-max rows: 0, 1, 0 (tie -> first), 1
    c = O.connectivity(h)
    # pairs in dist order: (1,2) (1,3) (1,4) (2,3) (2,4) (3,4)
    assert c.tolist() == [False, True, False, False, True, False]
    assert abs(O.dispersion(c, 4) - (1 / 4 + 8 * 6 * 0.25 / 16)) < 1e-15
    conav = np.array([0.9, 0.8, 0.1, 0.2, 0.15, 0.85])
    from scipy.cluster.hierarchy import cophenet, linkage
    from scipy.spatial.distance import squareform
    D = squareform(1 - conav)
    Z = linkage(squareform(D), "average")
    assert abs(O.cophenet(conav, 4) - cophenet(Z, squareform(D))[0]) < 1e-15
