"""GPU parity: the HIP engine, through the C ABI, against the CPU oracle on identical inputs.

Tolerances (fp64): one step on identical inputs -- factors
max relative error <= 1e-12, lkh relative error <= 1e-10; 50-step trajectories -- 1e-9.
The reference pins nothing for this path ("parity unpinned"): the oracle is the CPU
restatement of src/vbnmf_update.cpp, cross-checked against the R twin's restatement.
"""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

FACT = ("lw", "lh", "ew", "eh", "dw", "dh")
HY1 = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def check_step(got, want, tol_f=1e-12, tol_l=1e-10):
    for k in FACT:
        assert relerr(got[k], want[k]) <= tol_f, (k, relerr(got[k], want[k]))
    assert np.array_equal(got["w"], got["ew"]) and np.array_equal(got["h"], got["eh"])
    assert abs(got["lkh"] / want["lkh"] - 1) <= tol_l, (got["lkh"], want["lkh"])


def counts(n, m, lam, seed):
    rng = np.random.default_rng(seed)
    X = rng.poisson(lam, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1      # no empty rows
    X[rng.integers(0, n, m), np.arange(m)] += 1      # no empty columns
    return np.asfortranarray(X)


@pytest.mark.parametrize("n,m,r,lam", [(64, 96, 4, 1.5), (200, 500, 3, 0.8), (300, 400, 10, 0.05),
                                       (130, 70, 1, 1.0), (97, 211, 7, 0.3), (50, 60, 20, 2.0),
                                       (40, 45, 32, 1.0), (1500, 2300, 5, 0.1)])
def test_single_step_dense_matches_oracle(n, m, r, lam):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    X = counts(n, m, lam, seed=n + m + r)
    wh = synth.random_state(n, m, r, HY1, seed=7)
    got = C.vbnmf_update(X, wh, HY1, C.EPS)
    want = O.update_dense(X, wh, HY1, C.EPS)
    check_step(got, want)


def test_single_step_sparse_input_equals_dense_input():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = counts(120, 340, 0.2, seed=3)
    wh = synth.random_state(120, 340, 6, HY1, seed=8)
    a = C.vbnmf_update(X, wh, HY1, C.EPS)
    b = C.vbnmf_update(sp.csc_matrix(X), wh, HY1, C.EPS)
    c = C.vbnmf_update(sp.csr_matrix(X), wh, HY1, C.EPS)
    for k in FACT:
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k])
    assert a["lkh"] == b["lkh"] == c["lkh"]


def test_non_integer_counts_use_wide_layout():
    """normalize_count (reference R/utils.R:318-327) makes X non-integer: lgamma(X+1) matters."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    X = counts(80, 150, 0.7, seed=5)
    X = X * (np.median(X.sum(axis=0)) / X.sum(axis=0))[None, :]
    wh = synth.random_state(80, 150, 5, HY1, seed=9)
    check_step(C.vbnmf_update(X, wh, HY1, C.EPS), O.update_dense(X, wh, HY1, C.EPS))


def test_large_counts_above_the_packed_range_are_split():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    X = counts(40, 60, 1.0, seed=6)
    X[3, 4] = 70000.0
    wh = synth.random_state(40, 60, 3, HY1, seed=10)
    check_step(C.vbnmf_update(X, wh, HY1, C.EPS), O.update_dense(X, wh, HY1, C.EPS))


@pytest.mark.parametrize("hyper,fudge", [({"aw": 0.05, "bw": 1.0, "ah": 0.05, "bh": 1.0}, 2.220446049250313e-16),
                                         ({"aw": 2.5, "bw": 0.7, "ah": 0.3, "bh": 3.0}, 2.220446049250313e-16),
                                         ({"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}, 0.0),
                                         ({"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}, 1e-3)])
def test_hyper_and_fudge_variants(hyper, fudge):
    """Small shapes underflow exp(psi(a)) and hit the fudge floor (src/vbnmf_update.cpp:60,64)."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    X = counts(90, 120, 0.3, seed=11)
    wh = synth.random_state(90, 120, 4, hyper, seed=12)
    wh = {k: np.maximum(v, 1e-300) for k, v in wh.items()}
    check_step(C.vbnmf_update(X, wh, hyper, fudge), O.update_dense(X, wh, hyper, fudge))


def test_trajectory_fixed_hyper_50_steps():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    X = synth.drop_empty(synth.simulate_data(200, (100, 150, 250), seed=1, sparse=False))
    n, m = X.shape
    wh = synth.random_state(n, m, 3, HY1, seed=1001)
    eng = C.VBEngine(C.CountMatrix(X), 3)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    ref = wh
    for t in range(50):
        lkh, stats = eng.step(HY1)
        ref = O.update_dense(X, ref, HY1, C.EPS)
        assert abs(lkh / ref["lkh"] - 1) <= 1e-9, (t, lkh, ref["lkh"])
        want = (np.mean(np.log(ref["lw"])), np.mean(np.log(ref["lh"])), np.mean(ref["ew"]), np.mean(ref["eh"]))
        assert np.allclose(stats, want, rtol=1e-9, atol=0)
    got = eng.get_state()
    for k in FACT:
        assert relerr(got[k], ref[k]) <= 1e-9, k
    eng.close()


def test_trajectory_with_hyper_update_matches_oracle_loop():
    """vb_iterate semantics (reference R/bayesian.R:336-352): hyper_update from step n0+1 on,
    the lk0 lag on the convergence break, identical iteration count."""
    import ccfindr_amd as C
    from ccfindr_amd import synth, bayesian
    from oracle import vbnmf_oracle as O
    X = synth.drop_empty(synth.simulate_data(150, (80, 120), seed=4, sparse=False))
    n, m = X.shape
    r = 2
    wh0 = synth.random_state(n, m, r, HY1, seed=77)
    # oracle loop
    upd = lambda wh, hy, fud: O.update_dense(X, wh, hy, fud)
    whO, hyO, lk0O, itO, trace = O.vb_iterate(upd, dict(wh0), dict(HY1), Itmax=60, Tol=1e-5)
    # engine loop, written as the product's vb_iterate does it
    eng = C.VBEngine(C.CountMatrix(X), r)
    eng.set_state(wh0["lw"], wh0["lh"], wh0["eh"])
    hyper, lk0, it = dict(HY1), 0.0, 0
    for it in range(1, 61):
        lkh, stats = eng.step(hyper)
        if it > 10:
            hyper = bayesian.hyper_update((True,) * 4, stats, hyper, Niter=100, Tol=1e-3)
        assert abs(lkh / trace[it - 1][0] - 1) <= 1e-9, (it, lkh, trace[it - 1][0])
        for k in ("aw", "bw", "ah", "bh"):
            assert abs(hyper[k] / trace[it - 1][1][k] - 1) <= 1e-9
        if it > 1 and it > 10 and lkh >= lk0 and abs(1 - lkh / lk0) < 1e-5:
            break
        lk0 = lkh
    assert it == itO
    assert abs(lk0 / lk0O - 1) <= 1e-9
    eng.close()


def test_bit_reproducible_run_to_run():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = counts(400, 900, 0.1, seed=21)
    wh = synth.random_state(400, 900, 8, HY1, seed=22)
    outs = []
    for _ in range(2):
        eng = C.VBEngine(C.CountMatrix(X), 8)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        l = [eng.step(HY1)[0] for _ in range(5)]
        outs.append((l, eng.get_state()))
        eng.close()
    assert outs[0][0] == outs[1][0]
    for k in FACT:
        assert np.array_equal(outs[0][1][k], outs[1][1][k])


def test_algebraic_invariants():
    """sum_k sw_ik = rowSum(X)_i and sum_k sh_kj = colSum(X)_j (src/vbnmf_update.cpp:35-36), so
    sum_k alw_ik = r*aw + rowSum(X)_i; with ew = alw/bew and bew constant per k:
    sum_k ew_ik * bew_k = r*aw + rowSum(X)_i."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = counts(70, 110, 0.5, seed=31)
    r = 5
    wh = synth.random_state(70, 110, r, HY1, seed=32)
    got = C.vbnmf_update(X, wh, HY1, C.EPS)
    bew = 1.0 + wh["eh"].sum(axis=1)
    assert np.allclose((got["ew"] * bew[None, :]).sum(axis=1), r * 1.0 + X.sum(axis=1), rtol=1e-12)
    beh = 1.0 + got["ew"].sum(axis=0)
    assert np.allclose((got["eh"] * beh[:, None]).sum(axis=0), r * 1.0 + X.sum(axis=0), rtol=1e-12)
    assert np.allclose(got["dw"], got["ew"] / bew[None, :], rtol=1e-14)


def test_errors_are_statuses_not_crashes():
    import ccfindr_amd as C
    X = counts(20, 30, 1.0, seed=41)
    M = C.CountMatrix(X)
    with pytest.raises(C.VBNMFError) as ei:
        C.VBEngine(M, 0)
    assert ei.value.code == 1
    with pytest.raises(C.VBNMFError):
        C.VBEngine(M, C.MAX_RANK + 1)
    eng = C.VBEngine(M, 2)
    with pytest.raises(C.VBNMFError) as ei:
        eng.step(HY1)                       # step before set_state
    assert ei.value.code == 5
    eng.close()


def test_nan_propagates_to_lkh():
    """NaN is not an error: it reaches lkh so the caller's is.na() break works (R/bayesian.R:345)."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = counts(30, 40, 1.0, seed=51)
    wh = synth.random_state(30, 40, 2, HY1, seed=52)
    wh["lw"][3, 1] = np.nan
    got = C.vbnmf_update(X, wh, HY1, C.EPS)
    assert np.isnan(got["lkh"])


def test_stateless_entries_keep_the_ingested_matrix_between_calls():
    """vbnmf_update_dense / _csc are called with the same X on every iteration of the reference's loop
    (R/bayesian.R:339): the library keeps the last matrix and engine, keyed by X's content.  A repeat must give the
    same bits, a changed X the changed answer, and the repeat must be cheaper than the first call."""
    import time
    import scipy.sparse as sp
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    rng = np.random.default_rng(12)
    n, m, r = 600, 900, 5
    X = rng.poisson(0.5, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1
    X[rng.integers(0, n, m), np.arange(m)] += 1
    X = np.asfortranarray(X)
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    wh = synth.random_state(n, m, r, hy, seed=1)
    C.load().vbnmf_stateless_cache_clear()
    t0 = time.perf_counter(); a = C.vbnmf_update(X, wh, hy, C.EPS); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); b = C.vbnmf_update(X, wh, hy, C.EPS); t2 = time.perf_counter() - t0
    for k in ("lw", "lh", "ew", "eh", "dw", "dh"):
        assert np.array_equal(a[k], b[k]), k
    assert a["lkh"] == b["lkh"] and t2 < t1
    # chained calls, as vb_iterate makes them, against the oracle
    ref, cur = wh, wh
    for _ in range(3):
        cur = C.vbnmf_update(X, cur, hy, C.EPS)
        ref = O.update_dense(X, ref, hy, C.EPS)
        assert abs(cur["lkh"] / ref["lkh"] - 1) <= 1e-10
    # one changed entry: not the cached matrix any more
    X2 = X.copy(order="F"); X2[3, 4] += 2.0
    c = C.vbnmf_update(X2, wh, hy, C.EPS)
    want = O.update_dense(X2, wh, hy, C.EPS)
    assert abs(c["lkh"] / want["lkh"] - 1) <= 1e-10 and c["lkh"] != a["lkh"]
    # another rank on the same matrix, then the sparse entry
    wh7 = synth.random_state(n, m, 7, hy, seed=2)
    d = C.vbnmf_update(X2, wh7, hy, C.EPS)
    assert abs(d["lkh"] / O.update_dense(X2, wh7, hy, C.EPS)["lkh"] - 1) <= 1e-10
    S = sp.csc_matrix(X)
    e1 = C.vbnmf_update(S, wh, hy, C.EPS); e2 = C.vbnmf_update(S, wh, hy, C.EPS)
    assert e1["lkh"] == e2["lkh"] == a["lkh"] and np.array_equal(e1["ew"], a["ew"])
    C.load().vbnmf_stateless_cache_clear()
