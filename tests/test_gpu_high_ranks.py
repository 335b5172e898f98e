"""Ranks above 32 (reference limit: rank <= min(n, m) only, R/bayesian.R:319-320): the sweep's lanes share a task's
columns two by two (padded ranks 40, 48, 56, 64; kernels.h sweep_side SP = 2) or four by four (ranks 65..128, padded ranks
80, 96, 112, 128; SP = 4).  VB step, ML step, resident loop,
sparse product and the partitioned step against the oracles, on every padded rank and both layouts (packed / wide)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HY = {"aw": 1.2, "bw": 0.9, "ah": 0.8, "bh": 1.5}


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def counts(n, m, seed, kind="counts"):
    rng = np.random.default_rng(seed)
    X = rng.poisson(0.6, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1.0
    X[rng.integers(0, n, m), np.arange(m)] += 1.0
    if kind == "noninteger":                                   # wide layout (value + index streams)
        X = X * rng.uniform(0.5, 1.5, size=(1, m))
    elif kind == "binary":                                     # every stored value is 1: the leading-ones stretch
        X = (X > 0).astype(np.float64)
    return np.asfortranarray(X)


@pytest.mark.parametrize("kind", ["counts", "noninteger", "binary"])
@pytest.mark.parametrize("r", [33, 40, 47, 50, 56, 64, 65, 80, 90, 100, 112, 128])
def test_vb_step_above_rank_32(r, kind):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    n, m = 150, 230
    X = counts(n, m, r, kind)
    wh = synth.random_state(n, m, r, HY, seed=r)
    M = C.CountMatrix(X)
    eng = C.VBEngine(M, r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    cur = dict(wh)
    for step in range(3):                                      # resident steps: the fused evidence of step t + 1 too
        lkh, st = eng.step(HY)
        want = O.update_dense(X, cur, HY, C.EPS)
        assert abs(lkh / want["lkh"] - 1) <= 1e-10, (r, kind, step, lkh, want["lkh"])
        cur = want
    got = eng.get_state()
    for k in ("lw", "lh", "ew", "eh", "dw", "dh"):
        assert relerr(got[k], cur[k]) <= 1e-11, (r, kind, k, relerr(got[k], cur[k]))
    eng.close()


@pytest.mark.parametrize("r", [36, 64, 77, 128])
def test_stateless_entry_and_margins_above_rank_32(r):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    n, m = 90, 400
    X = counts(n, m, 7 + r)
    wh = synth.random_state(n, m, r, HY, seed=5)
    got = C.vbnmf_update(X, wh, HY, C.EPS)
    want = O.update_dense(X, wh, HY, C.EPS)
    for k in ("lw", "lh", "ew", "eh", "dw", "dh"):
        assert relerr(got[k], want[k]) <= 1e-12, (r, k)
    assert abs(got["lkh"] / want["lkh"] - 1) <= 1e-10
    # margin identity (SURVEY 8c KAT 3): sum_k alw_ik = r aw + rowSum(X)_i, with alw = ew * bew
    bew = HY["aw"] / HY["bw"] + wh["eh"].sum(axis=1)
    alw = got["ew"] * bew[None, :]
    assert np.allclose(alw.sum(axis=1), r * HY["aw"] + X.sum(axis=1), rtol=1e-12)


@pytest.mark.parametrize("r", [40, 64, 72, 128])
def test_ml_step_above_rank_32(r):
    import ccfindr_amd as C
    from oracle import mlnmf_oracle as O
    rng = np.random.default_rng(r)
    n, m = 120, 260
    X = counts(n, m, 3 * r)
    w, h = rng.uniform(0.05, 1.0, size=(n, r)), rng.uniform(0.05, 1.0, size=(r, m))
    got = C.nmf_update(X, w, h)
    want = O.nmf_update_literal(X, w, h)
    assert relerr(got["ew"], want["ew"]) <= 1e-12 and relerr(got["eh"], want["eh"]) <= 1e-12
    lk = O.likelihood_literal(X, want["ew"], want["eh"])
    wh = want["ew"] @ want["eh"]
    scale = (np.abs(X * np.log(wh)).sum() + wh.sum()) / n / m
    assert abs(got["lk"] - lk) <= 1e-11 * scale


@pytest.mark.parametrize("r", [48, 96])
def test_device_loop_above_rank_32_against_the_oracle_steps(r):
    """The resident loop (the control step: evidence, hyper-parameter Newton, stop rule) at a shared rank: its history of
    the first steps against the oracle stepped with the same hyper-parameter updates on the host."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from ccfindr_amd.bayesian import hyper_update
    from oracle import vbnmf_oracle as O
    n, m = 140, 300
    hy0 = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    X = counts(n, m, 11)
    M = C.CountMatrix(X)
    wh = synth.random_state(n, m, r, hy0, seed=2)
    eng = C.VBEngine(M, r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    out = eng.run(hy0, Itmax=12, Tol=0.0, n0=3, dn=1, history=True)
    assert out["it"] == 12 and out["reason"] == 4
    cur, hy = dict(wh), dict(hy0)
    for it in range(1, 13):
        cur = O.update_dense(X, cur, hy, C.EPS)
        assert abs(out["history"][it - 1, 0] / cur["lkh"] - 1) <= 1e-9, it
        if it > 3:
            hy = hyper_update((True,) * 4, cur, hy, Niter=100, Tol=1e-3)
    for k, v in hy.items():
        assert abs(out["hyper"][k] / v - 1) <= 1e-8, k
    eng.close()


@pytest.mark.parametrize("r", [40, 80])
def test_partitions_above_rank_32_equal_whole(r):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    n, m, cut = 100, 210, 90
    X = counts(n, m, 13)
    M = C.CountMatrix(X)
    wh = synth.random_state(n, m, r, HY, seed=3)
    whole = C.VBEngine(M, r)
    whole.set_state(wh["lw"], wh["lh"], wh["eh"])
    parts = [C.VBEngine(M, r, cols=(0, cut), m_global=m), C.VBEngine(M, r, cols=(cut, m), m_global=m)]
    comm = C.Communicator.local(2)
    for p, (b, e) in zip(parts, ((0, cut), (cut, m))):
        p.attach_comm(comm)
        p.set_state(wh["lw"], wh["lh"][:, b:e], wh["eh"][:, b:e])
    comm.state_finish()
    kw = dict(Itmax=9, Tol=0.0, n0=3, dn=1, history=True)
    want = whole.run(HY, **kw)
    got = comm.run(HY, **kw)                                   # the partitioned device loop: split sweep + group sum
    assert got["it"] == want["it"] == 9
    assert relerr(got["history"], want["history"]) <= 1e-10
    ref = whole.get_state()
    a, b = parts[0].get_state(), parts[1].get_state()
    assert np.array_equal(a["lw"], b["lw"])                    # gene side replicated bit for bit
    assert relerr(a["lw"], ref["lw"]) <= 1e-10
    assert relerr(np.concatenate([a["lh"], b["lh"]], axis=1), ref["lh"]) <= 1e-10
    comm.close()
    for e in parts + [whole]:
        e.close()


def test_sparse_product_and_truncated_svd_above_rank_32():
    import ccfindr_amd as C
    from ccfindr_amd import linalg
    n, m, k = 130, 240, 40
    X = counts(n, m, 17)
    M = C.CountMatrix(X)
    rng = np.random.default_rng(0)
    B = rng.standard_normal((k, m))
    eng = C.VBEngine(M, k)
    got = eng.spmm(B, transpose=False)                                    # X t(B): n x k
    assert np.allclose(got, X @ B.T, rtol=1e-12, atol=1e-10)
    Bt = rng.standard_normal((n, k))
    got_t = eng.spmm(Bt, transpose=True)                                  # t(B) X: k x m
    assert np.allclose(got_t, Bt.T @ X, rtol=1e-12, atol=1e-10)
    eng.close()
    for method in ("device", "host_qr"):
        U, s, Vt = linalg.truncated_svd(M, 36, method=method)             # engine rank 46 -> padded 48
        s_ref = np.linalg.svd(X, compute_uv=False)[:36]
        assert np.allclose(s[:8], s_ref[:8], rtol=1e-6), method
        assert np.allclose(U.T @ U, np.eye(U.shape[1]), atol=1e-8), method


def test_sparse_product_and_svd_above_64_columns():
    """Engine ranks above 64 (four lanes per task): the sparse products are there, the device-resident SVD is not (its
    k x k kernels hold 64 columns: a distinct error), and truncated_svd takes the host-QR form by itself."""
    import ccfindr_amd as C
    from ccfindr_amd import linalg
    n, m, k = 150, 260, 100
    X = counts(n, m, 23)
    M = C.CountMatrix(X)
    rng = np.random.default_rng(1)
    eng = C.VBEngine(M, k)
    B = rng.standard_normal((k, m))
    assert np.allclose(eng.spmm(B, transpose=False), X @ B.T, rtol=1e-12, atol=1e-10)
    Bt = rng.standard_normal((n, k))
    assert np.allclose(eng.spmm(Bt, transpose=True), Bt.T @ X, rtol=1e-12, atol=1e-10)
    with pytest.raises(C.VBNMFError) as ei:
        eng.svd(5)
    assert ei.value.code == 1 and "64" in str(ei.value)
    eng.close()
    U, s, Vt = linalg.truncated_svd(M, 70)                                # k = 80 columns: host QR over the GPU's products
    s_ref = np.linalg.svd(X, compute_uv=False)[:70]
    assert np.allclose(s[:8], s_ref[:8], rtol=1e-6)
    assert np.allclose(U.T @ U, np.eye(70), atol=1e-8)
    M.close()
