"""Both posterior updates in one launch (csrc/kernels.h: k_update2, the default of unpartitioned engines since round 5)
against the two-launch form of rounds 1-4 (VBNMF_NO_UPDATE_PAIR=1, read when an engine is created).

The reference updates `ew` before it forms `beh` from the NEW ew's column sums (src/vbnmf_update.cpp:44 before :53).  The
pair keeps that order without a kernel boundary: the gene side of the sweep leaves sum_i sw_ik, and
colSums(ew_new)_k = (n aw + sum_i sw_ik) / bew_k.  So:
  * the gene side (lw, ew, dw) must be the two-launch form's BIT FOR BIT (same gather, same rate);
  * the cell side differs by the rounding of one sum per column (a sum of n quotients against the quotient of a sum):
    held to 1e-13 relative after one step, far inside the 1e-12 of SURVEY.md section 8(c);
  * the loop (iteration count, stop reason, history) follows within the trajectory tolerance;
  * results are bit-reproducible run to run (no floating-point atomics: per-slice sums by a fixed tree, added in list order).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HY = {"aw": 1.1, "bw": 0.9, "ah": 0.8, "bh": 1.3}


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def _engine(M, r, pair, **kw):
    import ccfindr_amd as C
    keep = {k: os.environ.get(k) for k in ("VBNMF_NO_UPDATE_PAIR", "VBNMF_UPDATE_PAIR")}
    os.environ["VBNMF_NO_UPDATE_PAIR"] = "0" if pair else "1"
    os.environ["VBNMF_UPDATE_PAIR"] = "1" if pair else "0"       # (forced on: the default only takes it where it pays)
    try:
        return C.VBEngine(M, r, **kw)
    finally:
        for k, v in keep.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _matrix(kind, n, m, seed):
    from ccfindr_amd import synth
    rng = np.random.default_rng(seed)
    if kind == "clustered":
        X = synth.fill_empty(synth.simulate_data(n, [m // 3, m - m // 3], alpha0=0.2, seed=seed, depth=np.full(m, 80)))
        return X
    X = rng.poisson(0.6, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1.0
    X[rng.integers(0, n, m), np.arange(m)] += 1.0
    if kind == "noninteger":                                   # the wide layout (value + index streams)
        X = X * rng.uniform(0.5, 1.5, size=(1, m))
    return np.asfortranarray(X)


CASES = [("counts", 150, 230, 3), ("counts", 310, 190, 10), ("noninteger", 120, 260, 7), ("clustered", 400, 650, 5),
         ("counts", 97, 131, 20), ("counts", 150, 230, 31), ("counts", 150, 230, 40), ("noninteger", 150, 230, 64),
         ("counts", 140, 200, 96), ("counts", 2, 3, 1), ("counts", 64, 64, 2)]


@pytest.mark.parametrize("kind,n,m,r", CASES)
def test_one_step_pair_against_two_launches_and_the_oracle(kind, n, m, r):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    X = _matrix(kind, n, m, 11 * r + n)
    n, m = X.shape
    wh = synth.random_state(n, m, r, HY, seed=r)
    M = C.CountMatrix(X)
    got = []
    for pair in (True, False):
        eng = _engine(M, r, pair)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        lk = [eng.step(HY)[0] for _ in range(3)]               # three resident steps: the sweep's column sums feed steps 2, 3
        got.append((lk, eng.get_state()))
        eng.close()
    M.close()
    (lk_p, st_p), (lk_s, st_s) = got
    # (the gene side of the FIRST step is bit-identical in both forms -- next test; by step 3 the cell side's last-bit
    # differences have fed back into both factors)
    for k in ("lw", "ew", "dw", "lh", "eh", "dh"):
        assert relerr(st_p[k], st_s[k]) <= 1e-12, (k, relerr(st_p[k], st_s[k]))
    for a, b in zip(lk_p, lk_s):
        assert abs(a / b - 1) <= 1e-12
    cur = dict(wh)
    Xd = np.asarray(X.todense()) if hasattr(X, "todense") else X
    for _ in range(3):
        cur = O.update_dense(np.asfortranarray(Xd, dtype=np.float64), cur, HY, C.EPS)
    assert abs(lk_p[-1] / cur["lkh"] - 1) <= 1e-10
    for k in ("lw", "lh", "ew", "eh", "dw", "dh"):
        assert relerr(st_p[k], cur[k]) <= 1e-11, (k, relerr(st_p[k], cur[k]))


@pytest.mark.parametrize("r", [4, 10, 48])
def test_first_step_gene_side_is_bit_identical_and_cell_side_within_one_rounding(r):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = _matrix("counts", 260, 330, 40 + r)
    n, m = X.shape
    wh = synth.random_state(n, m, r, HY, seed=3)
    M = C.CountMatrix(X)
    st = []
    for pair in (True, False):
        eng = _engine(M, r, pair)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        eng.step(HY)
        st.append(eng.get_state())
        eng.close()
    M.close()
    for k in ("lw", "ew", "dw"):
        assert np.array_equal(st[0][k], st[1][k]), k
    for k in ("lh", "eh", "dh"):
        assert relerr(st[0][k], st[1][k]) <= 1e-13, (k, relerr(st[0][k], st[1][k]))


def test_pair_is_bit_reproducible_run_to_run_and_across_engines():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = _matrix("clustered", 500, 900, 9)
    n, m = X.shape
    r = 6
    wh = synth.random_state(n, m, r, HY, seed=1)
    M = C.CountMatrix(X)
    outs = []
    for rep in range(3):
        eng = _engine(M, r, True)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        a = eng.run(HY, Itmax=40, Tol=0.0, n0=5, dn=2, history=True)
        outs.append((a, eng.get_state()))
        eng.close()
    M.close()
    for a, st in outs[1:]:
        assert np.array_equal(a["history"], outs[0][0]["history"])
        for k in st:
            assert np.array_equal(st[k], outs[0][1][k]), k


@pytest.mark.parametrize("flags", [(True,) * 4, (False,) * 4])
def test_device_loop_pair_against_two_launches(flags):
    """The whole loop of vb_iterate (R/bayesian.R:336-352) on the device: same iteration count and stop reason, history and
    state within the trajectory tolerance; then a host-stepped step and a second run from the state the first left."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = _matrix("clustered", 400, 650, 8)
    n, m = X.shape
    r = 5
    wh = synth.random_state(n, m, r, HY, seed=2)
    M = C.CountMatrix(X)
    got = []
    for pair in (True, False):
        eng = _engine(M, r, pair)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        a = eng.run(HY, Itmax=300, Tol=3e-4, n0=4, dn=1, flags=flags, history=True)
        s1 = eng.step(a["hyper"])
        b = eng.run(a["hyper"], Itmax=13, Tol=0.0, n0=2, dn=1, flags=flags, history=True)
        got.append((a, s1, b, eng.get_state()))
        eng.close()
    M.close()
    (a0, s0, b0, st0), (a1, s1, b1, st1) = got
    assert a0["it"] == a1["it"] and a0["reason"] == a1["reason"] == 2
    assert b0["it"] == b1["it"] == 13 and b0["reason"] == b1["reason"] == 4
    assert relerr(a0["history"], a1["history"]) <= 1e-9 and relerr(b0["history"], b1["history"]) <= 1e-9
    assert abs(s0[0] / s1[0] - 1) <= 1e-10
    for k in st0:
        assert relerr(st0[k], st1[k]) <= 1e-9, k


def test_nan_state_reaches_the_evidence_through_the_pair():
    """A whole factor row 0 with fudge = 0: X / wth is NaN there (src/vbnmf_update.cpp:34); the evidence must be NaN and the
    loop must break with reason 1 (R/bayesian.R:345) -- the sweep's column sums carry the NaN like any other sum."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = _matrix("counts", 200, 300, 5)
    n, m = X.shape
    r = 3
    wh = synth.random_state(n, m, r, HY, seed=6)
    wh["lw"][0, :] = 0.0
    M = C.CountMatrix(X)
    outs = []
    for pair in (True, False):
        eng = _engine(M, r, pair)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        outs.append(eng.run(HY, Itmax=20, Tol=1e-5, fudge=0.0, flags=(False,) * 4, history=True))
        eng.close()
    M.close()
    assert outs[0]["reason"] == outs[1]["reason"] and outs[0]["it"] == outs[1]["it"]
    assert np.array_equal(np.isnan(outs[0]["history"][:, 0]), np.isnan(outs[1]["history"][:, 0]))
