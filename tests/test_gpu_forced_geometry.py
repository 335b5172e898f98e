"""Kernel paths that the default geometry only reaches on matrices too big for a unit test, forced through the
layout's environment switches: more minor blocks than workgroups (a workgroup walks SEVERAL segments, each with its own
staged block and ticket list), tasks cut into many short pieces, very few workgroups, workgroup counts that are not a
multiple of 8 (no XCD remapping).  VB step, resident steps and the ML step against the oracles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HY = {"aw": 1.2, "bw": 0.9, "ah": 0.8, "bh": 1.5}


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def counts(n, m, seed, noninteger=False):
    rng = np.random.default_rng(seed)
    X = rng.poisson(0.5, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1.0
    X[rng.integers(0, n, m), np.arange(m)] += 1.0
    X[3, :] += 1.0                                             # a dense gene: many pieces per block
    if noninteger:
        X = X * rng.uniform(0.5, 1.5, size=(1, m))
    return np.asfortranarray(X)


@pytest.mark.parametrize("r,lds_kb,nwg,max_len,noninteger", [
    (3, 8, 5, 16, False),        # ~100-minor blocks, 5 workgroups: several segments per workgroup
    (10, 8, 3, 8, False),        # shortest tasks (two groups), 3 workgroups
    (10, 8, 7, 32, True),        # wide layout (value + index streams)
    (30, 16, 4, 16, False),      # one-row-buffer loop
    (40, 24, 6, 16, False),      # two lanes per task
    (64, 32, 9, 24, True),       # two lanes per task, wide layout
])
def test_many_segments_per_workgroup(monkeypatch, r, lds_kb, nwg, max_len, noninteger):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    from oracle import mlnmf_oracle as OM
    monkeypatch.setenv("VBNMF_LDS_KB", str(lds_kb))
    monkeypatch.setenv("VBNMF_NWG", str(nwg))
    monkeypatch.setenv("VBNMF_MAX_LEN", str(max_len))
    n, m = 260, 700
    X = counts(n, m, r + nwg, noninteger)
    M = C.CountMatrix(X)
    from util_layout import build_layout
    v = build_layout(M, 0, r)                                  # the same switches cut the host-side copy of the layout
    assert v["n_wg"] == nwg and v["n_blocks"] > nwg and v["n_segs"] > nwg, (v["n_wg"], v["n_blocks"], v["n_segs"])
    assert v["wide"] == int(noninteger) and v["max_len"] <= max(max_len, 4)
    eng = C.VBEngine(M, r)
    wh = synth.random_state(n, m, r, HY, seed=r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    cur = dict(wh)
    for step in range(3):
        lkh, _ = eng.step(HY)
        cur = O.update_dense(X, cur, HY, C.EPS)
        assert abs(lkh / cur["lkh"] - 1) <= 1e-10, (step, lkh, cur["lkh"])
    got = eng.get_state()
    for k in ("lw", "lh", "ew", "eh", "dw", "dh"):
        assert relerr(got[k], cur[k]) <= 1e-11, (k, relerr(got[k], cur[k]))
    rng = np.random.default_rng(1)
    w, h = rng.uniform(0.05, 1.0, size=(n, r)), rng.uniform(0.05, 1.0, size=(r, m))
    eng.ml_set_state(w, h)
    lk = eng.ml_step()
    ml = eng.ml_get_state()
    want = OM.nmf_update_literal(X, w, h)
    assert relerr(ml["ew"], want["ew"]) <= 1e-12 and relerr(ml["eh"], want["eh"]) <= 1e-12
    eng.close()


@pytest.mark.parametrize("r,lds_kb,nwg", [(6, 8, 5), (40, 24, 6)])
def test_partition_group_loop_with_many_segments(monkeypatch, r, lds_kb, nwg):
    """The partitioned device loop (split sweep, k_pack, tails, group sum, k_control) on the forced geometry, against
    the single engine's loop on the default geometry."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from ccfindr_amd.parallel import cell_partition
    n, m = 200, 620
    X = counts(n, m, 31 + r)
    M = C.CountMatrix(X)
    wh = synth.random_state(n, m, r, HY, seed=9)
    kw = dict(Itmax=14, Tol=0.0, n0=4, dn=1, history=True)
    whole = C.VBEngine(M, r)                                   # default geometry
    whole.set_state(wh["lw"], wh["lh"], wh["eh"])
    want = whole.run(HY, **kw)
    ref = whole.get_state()
    monkeypatch.setenv("VBNMF_LDS_KB", str(lds_kb))
    monkeypatch.setenv("VBNMF_NWG", str(nwg))
    monkeypatch.setenv("VBNMF_MAX_LEN", "16")
    cuts = cell_partition(m, 3)
    comm = C.Communicator.local(len(cuts))
    parts = [C.VBEngine(M, r, cols=c, m_global=m) for c in cuts]
    for p, (b, e) in zip(parts, cuts):
        p.attach_comm(comm)
        p.set_state(wh["lw"], wh["lh"][:, b:e], wh["eh"][:, b:e])
    comm.state_finish()
    got = comm.run(HY, **kw)
    assert got["it"] == want["it"] == 14
    assert relerr(got["history"], want["history"]) <= 1e-10
    st = [p.get_state() for p in parts]
    assert np.array_equal(st[0]["lw"], st[1]["lw"]) and np.array_equal(st[0]["lw"], st[2]["lw"])
    assert relerr(st[0]["lw"], ref["lw"]) <= 1e-9
    assert relerr(np.concatenate([q["lh"] for q in st], axis=1), ref["lh"]) <= 1e-9
    comm.close()
    for e in parts + [whole]:
        e.close()


@pytest.mark.parametrize("r,planned", [(3, (3, 20)), (10, (2, 10, 18)), (12, (12, 40)), (40, (40, 100)), (70, (70, 128))])
def test_step_under_a_rank_class_geometry_matches_the_oracle(r, planned):
    """An engine whose layouts were cut for a LARGER rank of the same sweep (CountMatrix.plan_ranks: wider LDS row stride,
    narrower blocks) computes the same step: factors 1e-12, evidence 1e-10 against the oracle."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    X = synth.fill_empty(synth.simulate_data(900, [700] * 3, alpha0=0.1, seed=13, depth=np.full(2100, 150)))
    n, m = X.shape
    M = C.CountMatrix(X)
    M.plan_ranks(planned)
    wh = synth.random_state(n, m, r, hy, seed=r)
    eng = C.VBEngine(M, r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    lkh, _ = eng.step(hy)
    got = eng.get_state()
    eng.close()
    M.close()
    want = O.update_csc(n, m, X.indptr, X.indices, X.data, wh, hy, nthreads=8)
    assert abs(lkh / want["lkh"] - 1) <= 1e-10
    for k in ("lw", "lh", "ew", "eh", "dw", "dh"):
        err = float(np.max(np.abs(got[k] - want[k]) / np.abs(want[k])))
        assert err <= 1e-12, (k, err)
