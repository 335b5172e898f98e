"""The layout's internal renumbering of the cells (csrc/order.cpp) is invisible at every boundary.  An engine whose
layouts store the cells in a clustered order (VBNMF_CELL_ORDER=1 forces it for any size; it is automatic from 8 192 cells)
must agree with one that stores them as given (=0) wherever cell-indexed data crosses the C ABI: set_state / get_state,
the device-driven loop's history, random_state's draws (keyed by the caller's element index), the ML step, arg-max labels
and their change count, the sparse products and the truncated SVD's right vectors, and a cell-partitioned group.
Summation orders differ between the two layouts, so floating-point results agree to the step tolerance (1e-12 / 1e-10), the
integer ones exactly.  Reference step: src/vbnmf_update.cpp:33-90."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HY = {"aw": 1.1, "bw": 0.9, "ah": 0.8, "bh": 1.3}


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def _matrix():
    from ccfindr_amd import synth
    # three cell types with distinct gene programmes, cells shuffled: something for the ordering to find
    return synth.fill_empty(synth.simulate_data(900, [500, 700, 600], alpha0=0.05, seed=31, depth=np.full(1800, 160)), seed=31)


class _Ordered:
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("VBNMF_CELL_ORDER")
        os.environ["VBNMF_CELL_ORDER"] = "1" if self.on else "0"

    def __exit__(self, *a):
        if self.old is None:
            del os.environ["VBNMF_CELL_ORDER"]
        else:
            os.environ["VBNMF_CELL_ORDER"] = self.old


def _pair(X, r, **kw):
    """(engine on an ordered layout, engine on an unordered one, their matrices)."""
    import ccfindr_amd as C
    out = []
    for on in (True, False):
        with _Ordered(on):
            M = C.CountMatrix(X)
            out.append((C.VBEngine(M, r, **kw), M))          # the order is fixed when the first layout is cut
    return out


def test_the_ordering_really_renumbers_and_cuts_fewer_gene_side_tasks():
    from util_layout import build_layout
    import ccfindr_amd as C
    X = _matrix()
    tasks = {}
    for on in (True, False):
        with _Ordered(on):
            M = C.CountMatrix(X)
            v = build_layout(M, 0, 20)
            tasks[on] = v["n_tasks"]
            assert (v["cell_perm"] is not None) == on
            if on:
                w = build_layout(M, 1, 20)
                assert np.array_equal(v["cell_perm"], w["cell_perm"])           # both sides in the same order
                assert sorted(v["cell_perm"].tolist()) == list(range(X.shape[1]))
            M.close()
    assert tasks[True] < tasks[False]


def test_state_loop_and_labels_agree_with_the_unordered_layout():
    from ccfindr_amd import synth
    X = _matrix()
    n, m = X.shape
    r = 5
    wh = synth.random_state(n, m, r, HY, seed=2)
    (a, Ma), (b, Mb) = _pair(X, r)
    res = []
    for eng in (a, b):
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        st0 = eng.get_state(("lw", "lh", "eh"))
        assert np.array_equal(st0["lh"], wh["lh"]) and np.array_equal(st0["eh"], wh["eh"])      # a pure round trip
        out = eng.run(HY, Itmax=25, Tol=0.0, n0=5, dn=1, flags=(True,) * 4, history=True)
        ch0, ids0 = eng.cluster_changes(want_ids=True)
        lkh, stats = eng.step(out["hyper"])
        ch1, ids1 = eng.cluster_changes(want_ids=True)
        res.append((out, eng.get_state(), eng.cluster_ids(), ch0, ids0, ch1, ids1, lkh))
    (oa, sa, ia, c0a, i0a, c1a, i1a, la), (ob, sb, ib, c0b, i0b, c1b, i1b, lb) = res
    assert relerr(oa["history"], ob["history"]) <= 1e-10 and abs(la / lb - 1) <= 1e-10
    for k in sa:
        assert relerr(sa[k], sb[k]) <= 1e-9, k
    assert np.array_equal(ia, ib) and np.array_equal(i0a, i0b) and np.array_equal(i1a, i1b)     # labels in the CALLER's cell order
    assert c0a is None and c0b is None and c1a == c1b
    for eng, M in ((a, Ma), (b, Mb)):
        eng.close(); M.close()


def test_random_state_draws_do_not_depend_on_the_order():
    X = _matrix()
    (a, Ma), (b, Mb) = _pair(X, 4)
    for eng in (a, b):
        eng.random_state(HY, seed=12345)
    sa, sb = a.get_state(("lw", "lh", "eh")), b.get_state(("lw", "lh", "eh"))
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k               # keyed by (seed, factor, caller's element index): bit-identical
    for eng, M in ((a, Ma), (b, Mb)):
        eng.close(); M.close()


def test_ml_step_products_and_svd_agree():
    X = _matrix()
    n, m = X.shape
    r = 6
    rng = np.random.default_rng(5)
    w0, h0 = rng.uniform(size=(n, r)), rng.uniform(size=(r, m))
    B_cells, B_genes = rng.normal(size=(r, m)), rng.normal(size=(n, r))
    (a, Ma), (b, Mb) = _pair(X, r)
    outs = []
    for eng in (a, b):
        eng.ml_set_state(w0, h0)
        lk = [eng.ml_step() for _ in range(3)]
        st = eng.ml_get_state()
        xb = eng.spmm(B_cells)                               # X B^T : gathers one row of B per cell
        bx = eng.spmm(B_genes, transpose=True)               # B^T X : one column per cell out
        u, d, vt, _ = eng.svd(3, tol=1e-10, maxit=80, seed=3)
        outs.append((lk, st, xb, bx, d, vt))
    (la, sa, xa, ba, da, va), (lb, sb, xb_, bb, db, vb) = outs
    assert relerr(np.array(la), np.array(lb)) <= 1e-10
    for k in sa:
        assert relerr(sa[k], sb[k]) <= 1e-10, k
    D = X.toarray()
    want_xb, want_bx = D @ B_cells.T, B_genes.T @ D
    for got in (xa, xb_):
        assert np.max(np.abs(got - want_xb)) <= 1e-11 * np.max(np.abs(want_xb))
    for got in (ba, bb):
        assert np.max(np.abs(got - want_bx)) <= 1e-11 * np.max(np.abs(want_bx))
    assert relerr(da, db) <= 1e-8
    for q in range(3):                                       # right vectors up to sign, columns in the caller's cell order
        s = np.sign(np.dot(va[q], vb[q]))
        assert np.max(np.abs(va[q] - s * vb[q])) <= 1e-5 * np.max(np.abs(vb[q]))
    for eng, M in ((a, Ma), (b, Mb)):
        eng.close(); M.close()


def test_partition_engines_order_their_own_cells():
    """Every partition renumbers ITS cells (both of its sides alike); the group must still reproduce the single engine."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from ccfindr_amd.parallel import cell_partition
    X = _matrix()
    n, m = X.shape
    r, P = 4, 3
    wh = synth.random_state(n, m, r, HY, seed=8)
    kw = dict(Itmax=21, Tol=0.0, n0=5, dn=1, flags=(True,) * 4, history=True)
    with _Ordered(False):
        M0 = C.CountMatrix(X)
        whole = C.VBEngine(M0, r)
    whole.set_state(wh["lw"], wh["lh"], wh["eh"])
    want = whole.run(HY, **kw)
    ref = whole.get_state()
    with _Ordered(True):
        M = C.CountMatrix(X)
        cuts = cell_partition(m, P)
        comm = C.Communicator.local(P)
        parts = [C.VBEngine(M, r, cols=c, m_global=m) for c in cuts]
    for p, (b, e) in zip(parts, cuts):
        p.attach_comm(comm)
        p.set_state(wh["lw"], wh["lh"][:, b:e], wh["eh"][:, b:e])
    comm.state_finish()
    got = comm.run(HY, **kw)
    assert got["it"] == want["it"] and relerr(got["history"], want["history"]) <= 1e-10
    st = [p.get_state() for p in parts]
    assert relerr(st[0]["ew"], ref["ew"]) <= 1e-9
    assert relerr(np.concatenate([q["eh"] for q in st], axis=1), ref["eh"]) <= 1e-9
    for e in parts + [whole]:
        e.close()
    comm.close(); M.close(); M0.close()
