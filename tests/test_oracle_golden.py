"""The CPU oracle against the committed golden vectors, and its known-answer tests.

PARITY UNPINNED (see oracle/vbnmf_oracle.c): the goldens pin the oracle, which is the literal
restatement of reference src/vbnmf_update.cpp; they cannot pin the reference, which holds no
vectors for this path.  What is checked here: (1) the C restatement still reproduces the
committed vectors on this machine, (2) the stored-entries (CSC) form and the numpy restatement
of the R twin (R/bayesian.R:56-106) agree with them, (3) algebraic known answers that need no
oracle at all.
"""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import vbnmf_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FACT = ("lw", "lh", "ew", "eh", "dw", "dh")


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def load_step(path):
    z = np.load(path)
    hy = dict(zip(("aw", "bw", "ah", "bh"), (float(v) for v in z["hyper"])))
    wh = {"lw": z["lw0"], "lh": z["lh0"], "eh": z["eh0"]}
    return z, z["X"], wh, hy, float(z["fudge"])


STEP_FILES = sorted(glob.glob(os.path.join(GOLD, "step_*.npz")))


def test_fixtures_present():
    assert len(STEP_FILES) >= 7


@pytest.mark.parametrize("path", STEP_FILES, ids=lambda p: os.path.basename(p)[5:-4])
def test_dense_restatement_reproduces_golden(path):
    z, X, wh, hy, fudge = load_step(path)
    got = O.update_dense(X, wh, hy, fudge)
    for k in FACT:
        assert relerr(got[k], z[k]) <= 1e-13, k
    assert abs(got["lkh"] / float(z["lkh"]) - 1) <= 1e-13


@pytest.mark.parametrize("path", STEP_FILES, ids=lambda p: os.path.basename(p)[5:-4])
def test_csc_form_and_r_twin_agree_with_golden(path):
    z, X, wh, hy, fudge = load_step(path)
    S = sp.csc_matrix(X)
    n, m = X.shape
    for got in (O.update_csc(n, m, S.indptr, S.indices, S.data, wh, hy, fudge, nthreads=2),
                O.update_rtwin(X, wh, hy, fudge)):
        for k in FACT:
            assert relerr(got[k], z[k]) <= 1e-12, k
        assert abs(got["lkh"] / float(z["lkh"]) - 1) <= 1e-12


def test_pbmc_sample_golden():
    z = np.load(os.path.join(GOLD, "pbmc_extdata_r5.npz"))
    n, m = int(z["n"]), int(z["m"])
    wh = {"lw": z["lw0"], "lh": z["lh0"], "eh": z["eh0"]}
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    got = O.update_csc(n, m, z["indptr"], z["indices"], z["data"].astype(np.float64), wh, hy, nthreads=2)
    for k in FACT:
        assert relerr(got[k], z[k]) <= 1e-12, k
    assert abs(got["lkh"] / float(z["lkh"]) - 1) <= 1e-12


@pytest.mark.parametrize("tag", ["fixed", "hyper"])
def test_trajectory_golden(tag):
    z = np.load(os.path.join(GOLD, f"traj_{tag}_c1_r3.npz"))
    X = z["X"]
    wh = {"lw": z["lw0"], "lh": z["lh0"], "eh": z["eh0"], "ew": z["lw0"]}
    flags = (tag == "hyper",) * 4
    upd = lambda w, h, f: O.update_dense(X, w, h, f)
    whT, hy, lk0, it, trace = O.vb_iterate(upd, wh, {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}, Itmax=50, Tol=0.0,
                                           hyper_flags=flags)
    assert it == 50
    assert np.allclose([t[0] for t in trace], z["lkh"], rtol=1e-11, atol=0)
    assert np.allclose([[t[1][k] for k in ("aw", "bw", "ah", "bh")] for t in trace], z["hyper"], rtol=1e-10, atol=0)
    assert relerr(whT["ew"], z["ew"]) <= 1e-9 and relerr(whT["eh"], z["eh"]) <= 1e-9


def test_loop_golden_iteration_count_and_lk0_lag():
    """On a convergence break lk0 is NOT refreshed (R/bayesian.R:347 precedes :348)."""
    z = np.load(os.path.join(GOLD, "loop_c1_r3.npz"))
    X = z["X"]
    wh = {"lw": z["lw0"], "lh": z["lh0"], "eh": z["eh0"]}
    upd = lambda w, h, f: O.update_dense(X, w, h, f)
    whT, hy, lk0, it, trace = O.vb_iterate(upd, wh, {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}, Itmax=400, Tol=1e-5)
    assert it == int(z["it"]) and it < 400
    assert abs(lk0 / float(z["lk0"]) - 1) <= 1e-10
    assert lk0 == trace[-2][0] and trace[-1][0] != lk0          # reported evidence is the previous step's


# ---------------------------------------------------------------- algebraic known answers
def _small(seed=5, n=40, m=55, r=4):
    rng = np.random.default_rng(seed)
    X = rng.poisson(0.8, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1
    X[rng.integers(0, n, m), np.arange(m)] += 1
    hy = {"aw": 1.3, "bw": 0.8, "ah": 0.7, "bh": 1.9}
    wh = O.vb_init_random(n, m, r, hy, rng)
    return X, wh, hy


def test_kat_statistics_sum_to_margins():
    """sum_k sw_ik = rowSum(X)_i, sum_k sh_kj = colSum(X)_j (src/vbnmf_update.cpp:33-36)."""
    X, wh, hy = _small()
    out = O.update_dense(X, wh, hy)
    r = wh["lw"].shape[1]
    bew = hy["aw"] / hy["bw"] + wh["eh"].sum(axis=1)
    assert np.allclose((out["ew"] * bew[None, :]).sum(axis=1), r * hy["aw"] + X.sum(axis=1), rtol=1e-13)
    beh = hy["ah"] / hy["bh"] + out["ew"].sum(axis=0)
    assert np.allclose((out["eh"] * beh[:, None]).sum(axis=0), r * hy["ah"] + X.sum(axis=0), rtol=1e-13)
    assert np.allclose(out["dw"], out["ew"] / bew[None, :], rtol=1e-14)
    assert np.array_equal(out["w"], out["ew"]) and np.array_equal(out["h"], out["eh"])


def test_kat_rank_one_closed_form():
    X, wh, hy = _small(r=1)
    out = O.update_dense(X, wh, hy)
    bew = hy["aw"] / hy["bw"] + wh["eh"].sum()
    assert np.allclose(out["ew"][:, 0], (hy["aw"] + X.sum(axis=1)) / bew, rtol=1e-13)
    beh = hy["ah"] / hy["bh"] + out["ew"].sum()
    assert np.allclose(out["eh"][0], (hy["ah"] + X.sum(axis=0)) / beh, rtol=1e-13)


def test_kat_permutation_equivariance():
    X, wh, hy = _small()
    rng = np.random.default_rng(1)
    pr, pc, pk = rng.permutation(X.shape[0]), rng.permutation(X.shape[1]), rng.permutation(wh["lw"].shape[1])
    a = O.update_dense(X, wh, hy)
    whp = {"lw": wh["lw"][pr][:, pk], "lh": wh["lh"][pk][:, pc], "eh": wh["eh"][pk][:, pc]}
    b = O.update_dense(X[pr][:, pc], whp, hy)
    assert relerr(b["ew"], a["ew"][pr][:, pk]) <= 1e-12 and relerr(b["eh"], a["eh"][pk][:, pc]) <= 1e-12
    assert abs(a["lkh"] / b["lkh"] - 1) <= 1e-12


def test_kat_hyper_update_bh_is_always_overwritten():
    """R/bayesian.R:50-51: `bh1 <- ehm` in both branches."""
    X, wh, hy = _small()
    out = O.update_dense(X, wh, hy)
    new = O.hyper_update((True, True, True, False), out, hy)
    assert new["bh"] == pytest.approx(float(np.mean(out["eh"])), rel=1e-15)
    new2 = O.hyper_update((False, False, False, False), out, hy)
    assert new2 == hy                                            # all flags off: untouched (:4)


def test_special_functions_against_mpmath():
    import mpmath
    xs = np.concatenate([np.logspace(-8, 8, 200), np.linspace(1.40, 1.52, 41), [1.4616321449683623]])
    dg = O.digamma(xs)
    ref = np.array([float(mpmath.digamma(mpmath.mpf(float(x)))) for x in xs])
    assert np.max(np.abs(dg - ref) / np.maximum(1, np.abs(ref))) < 2e-15
    tg = O.trigamma(xs)
    ref = np.array([float(mpmath.polygamma(1, mpmath.mpf(float(x)))) for x in xs])
    assert np.max(np.abs(tg - ref) / np.abs(ref)) < 2e-15
