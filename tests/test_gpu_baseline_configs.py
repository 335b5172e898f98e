"""GPU parity at the sizes BASELINE.json names (configs 2 and 3), plus size-independent properties.

The oracle's stored-entries form finishes one step of the 20k x 50k matrix in about a second, so the
headline workload is checked directly against it, not only through invariants.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HY1 = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
FACT = ("lw", "lh", "ew", "eh", "dw", "dh")


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def test_config2_dense_2k_x_10k_rank5():
    """BASELINE config 2: 2000 x 10000 dense counts (~80 % non-zero), rank 5, fp64."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    X = synth.simulate_data(2000, [2000] * 5, alpha0=2.0, seed=2, depth=np.full(10000, 4000), sparse=False)
    X = synth.drop_empty(X)
    n, m = X.shape
    assert n == 2000 and m == 10000 and (X > 0).mean() > 0.7
    wh = synth.random_state(n, m, 5, HY1, seed=1002)
    got = C.vbnmf_update(X, wh, HY1, C.EPS)
    want = O.update_dense(X, wh, HY1, C.EPS)
    for k in FACT:
        assert relerr(got[k], want[k]) <= 1e-12, (k, relerr(got[k], want[k]))
    assert abs(got["lkh"] / want["lkh"] - 1) <= 1e-10


@pytest.fixture(scope="module")
def c3():
    import bench
    name, X, r = bench.make_workload(False)
    return X, r


def test_config3_headline_workload_against_sparse_oracle(c3):
    """BASELINE config 3 (the bench workload): three resident steps vs the oracle's CSC form."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    X, r = c3
    n, m = X.shape
    assert (n, m, r) == (20000, 50000, 10) and 0.048 < X.nnz / (n * m) < 0.052
    wh = synth.random_state(n, m, r, HY1, seed=1003)
    eng = C.VBEngine(C.CountMatrix(X), r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    ref = wh
    for _ in range(3):
        lkh, stats = eng.step(HY1)
        ref = O.update_csc(n, m, X.indptr, X.indices, X.data, ref, HY1, nthreads=16)
        assert abs(lkh / ref["lkh"] - 1) <= 1e-10
        assert np.allclose(stats, (np.mean(np.log(ref["lw"])), np.mean(np.log(ref["lh"])), np.mean(ref["ew"]), np.mean(ref["eh"])), rtol=1e-10)
    got = eng.get_state()
    for k in FACT:
        assert relerr(got[k], ref[k]) <= 1e-11, (k, relerr(got[k], ref[k]))
    eng.close()


def test_config3_size_independent_properties(c3):
    """sum_k sw_ik = rowSum(X)_i and sum_k sh_kj = colSum(X)_j (src/vbnmf_update.cpp:33-36), and
    runs are bit-reproducible."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X, r = c3
    n, m = X.shape
    wh = synth.random_state(n, m, r, HY1, seed=7)
    M = C.CountMatrix(X)
    runs = []
    for _ in range(2):
        eng = C.VBEngine(M, r)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        lk = [eng.step(HY1)[0] for _ in range(6)]
        runs.append((lk, eng.get_state(("ew", "eh"))))
        eng.close()
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1]["ew"], runs[1][1]["ew"])
    assert all(np.isfinite(v) for v in runs[0][0])
    # (the evidence need not rise monotonically: sw and sh are both formed from the OLD lw, lh,
    # src/vbnmf_update.cpp:33-36, so a step is not an exact coordinate ascent)
    # first step's margins, from a fresh engine
    eng = C.VBEngine(M, r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    eng.step(HY1)
    st = eng.get_state(("ew", "eh"))
    eng.close()
    bew = 1.0 + wh["eh"].sum(axis=1)
    rows = np.asarray(X.sum(axis=1)).ravel()
    assert np.allclose((st["ew"] * bew[None, :]).sum(axis=1), r * 1.0 + rows, rtol=1e-11)
    beh = 1.0 + st["ew"].sum(axis=0)
    cols = np.asarray(X.sum(axis=0)).ravel()
    assert np.allclose((st["eh"] * beh[:, None]).sum(axis=0), r * 1.0 + cols, rtol=1e-11)


@pytest.mark.parametrize("r", [2, 7, 12, 16, 20, 32])
def test_rank_sweep_ranks_on_a_mid_size_matrix(r):
    """BASELINE config 4's ranks (2..20) and the largest supported rank, one step each vs the oracle."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    X = synth.fill_empty(synth.simulate_data(1500, [800] * 4, alpha0=0.1, seed=9, depth=np.full(3200, 200)))
    n, m = X.shape
    wh = synth.random_state(n, m, r, HY1, seed=r)
    got = C.vbnmf_update(X, wh, HY1, C.EPS)
    want = O.update_csc(n, m, X.indptr, X.indices, X.data, wh, HY1, nthreads=8)
    for k in FACT:
        assert relerr(got[k], want[k]) <= 1e-12, (k, relerr(got[k], want[k]))
    assert abs(got["lkh"] / want["lkh"] - 1) <= 1e-10
