"""The device-driven loop of cell-partitioned engines (vbnmf_engine_run with a communicator, vbnmf_group_run):
per step the gene-side sweep, k_pack, the n x r all-reduce on a second stream beside the cell-side sweep, the small
all-reduce of the evidence partials, the control step folded into the next update -- all queued from C++ (reference loop: R/bayesian.R:337-352; exchange: SURVEY.md section 8e).

ONE test GPU, so the multi-partition runs use a local group (partition engines side by side in this process, the sum
a kernel in partition order) and the RCCL path runs with one rank; both must reproduce the single engine."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HY = {"aw": 1.2, "bw": 0.9, "ah": 0.8, "bh": 1.5}


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def _matrix(n, m, seed):
    rng = np.random.default_rng(seed)
    X = rng.poisson(0.3, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1
    X[rng.integers(0, n, m), np.arange(m)] += 1
    return X


def _group(M, r, cuts, m, wh):
    import ccfindr_amd as C
    comm = C.Communicator.local(len(cuts))
    parts = [C.VBEngine(M, r, cols=c, m_global=m) for c in cuts]
    for p, (b, e) in zip(parts, cuts):
        p.attach_comm(comm)
        p.set_state(wh["lw"], wh["lh"][:, b:e], wh["eh"][:, b:e])
    comm.state_finish()
    return comm, parts


@pytest.mark.parametrize("n,m,r,P,flags", [(300, 501, 6, 3, True), (120, 260, 4, 2, False), (200, 333, 11, 4, True)])
def test_local_group_device_loop_equals_single_engine(n, m, r, P, flags):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from ccfindr_amd.parallel import cell_partition
    X = _matrix(n, m, n + m)
    wh = synth.random_state(n, m, r, HY, seed=3)
    M = C.CountMatrix(X)
    kw = dict(Itmax=37, Tol=0.0, n0=10, dn=1, flags=(flags,) * 4, history=True)
    whole = C.VBEngine(M, r)
    whole.set_state(wh["lw"], wh["lh"], wh["eh"])
    want = whole.run(HY, **kw)
    ref = whole.get_state()
    cuts = cell_partition(m, P)
    comm, parts = _group(M, r, cuts, m, wh)
    got = comm.run(HY, **kw)
    assert got["it"] == want["it"] == 37 and got["reason"] == want["reason"] == 4
    assert relerr(got["history"], want["history"]) <= 1e-10
    assert abs(got["lkh"] / want["lkh"] - 1) <= 1e-11 and abs(got["lk0"] / want["lk0"] - 1) <= 1e-11
    for k in ("aw", "bw", "ah", "bh"):
        assert abs(got["hyper"][k] / want["hyper"][k] - 1) <= 1e-10
    st = [p.get_state() for p in parts]
    for k in ("lw", "ew", "dw"):
        for q in st[1:]:
            assert np.array_equal(st[0][k], q[k])                     # gene side replicated bit for bit
        assert relerr(st[0][k], ref[k]) <= 1e-9
    for k in ("lh", "eh", "dh"):
        assert relerr(np.concatenate([q[k] for q in st], axis=1), ref[k]) <= 1e-9
    # the host-stepped protocol goes on from the loop's state (reduced statistics are back in the reduce buffer)
    import torch
    reds = [p.reduce_tensor() for p in parts]
    for p in parts:
        p.step_local(got["hyper"])
    torch.cuda.synchronize()
    s = sum(reds[1:], reds[0].clone())
    for q in reds:
        q.copy_(s)
    torch.cuda.synchronize()
    outs = [p.step_finish() for p in parts]
    assert all(o == outs[0] for o in outs)
    lkh1, _ = whole.step(want["hyper"])
    assert abs(outs[0][0] / lkh1 - 1) <= 1e-10
    for e in parts + [whole]:
        e.close()
    comm.close()


def test_local_group_loop_stops_where_the_single_engine_stops():
    """A convergence break (R/bayesian.R:346-347) inside a queued batch: steps queued past it -- kernels AND
    all-reduces -- must leave the state exactly as the breaking step left it."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from ccfindr_amd.parallel import cell_partition
    n, m, r, P = 150, 400, 3, 3
    X = synth.drop_empty(synth.simulate_data(n, (120, 130, 150), seed=8, sparse=True))
    n, m = X.shape
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    wh = synth.random_state(n, m, r, hy, seed=5)
    M = C.CountMatrix(X)
    kw = dict(Itmax=2000, Tol=1e-5, n0=10, dn=1, flags=(True,) * 4)
    whole = C.VBEngine(M, r)
    whole.set_state(wh["lw"], wh["lh"], wh["eh"])
    want = whole.run(hy, **kw)
    assert want["reason"] == 2 and 12 < want["it"] < 2000
    comm, parts = _group(M, r, cell_partition(m, P), m, wh)
    got = comm.run(hy, **kw)
    assert got["it"] == want["it"] and got["reason"] == 2
    assert abs(got["lk0"] / want["lk0"] - 1) <= 1e-9 and abs(got["lkh"] / want["lkh"] - 1) <= 1e-9
    ref = whole.get_state(("ew", "eh"))
    st = [p.get_state(("ew", "eh")) for p in parts]
    assert relerr(st[0]["ew"], ref["ew"]) <= 1e-8
    assert relerr(np.concatenate([q["eh"] for q in st], axis=1), ref["eh"]) <= 1e-8
    for e in parts + [whole]:
        e.close()
    comm.close()


def test_rccl_communicator_one_rank_device_loop():
    """The RCCL form of the same loop with a 1-rank communicator: ncclAllReduce enqueued from C++ on the engine's comm
    stream.  The engine owns every cell but is declared one partition of a matrix twice as wide (the other partition
    would be empty), which only rescales the per-element evidence by 1/2."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    n, m, r = 220, 340, 5
    X = _matrix(n, m, 77)
    wh = synth.random_state(n, m, r, HY, seed=6)
    M = C.CountMatrix(X)
    kw = dict(Itmax=25, Tol=0.0, flags=(False,) * 4, history=True)
    whole = C.VBEngine(M, r)
    whole.set_state(wh["lw"], wh["lh"], wh["eh"])
    want = whole.run(HY, **kw)
    comm = C.Communicator.rccl(C.Communicator.unique_id(), 1, 0, 0)
    part = C.VBEngine(M, r, cols=(0, m), m_global=2 * m)
    part.attach_comm(comm)
    part.set_state(wh["lw"], wh["lh"], wh["eh"])
    part.allreduce()
    part.state_finish()
    got = part.run(HY, **kw)
    assert got["it"] == 25 and got["reason"] == 4
    assert relerr(2.0 * got["history"][:, 0], want["history"][:, 0]) <= 1e-10
    a, b = part.get_state(), whole.get_state()
    for k in a:
        assert relerr(a[k], b[k]) <= 1e-9, k
    # host-stepped with the library's all-reduce
    part.step_local(HY); part.allreduce()
    lkh, _ = part.step_finish()
    assert abs(2.0 * lkh / whole.step(HY)[0] - 1) <= 1e-10
    part.close(); whole.close(); comm.close()


@pytest.mark.parametrize("tol,itmax", [(0.0, 23), (2e-4, 400), (0.0, 1)])
def test_partitioned_control_fold_equals_the_separate_control_kernel(tol, itmax, monkeypatch):
    """Partitioned engines fold the control step into the next step's gene-side update too (the second exchange of a step
    carries the sweeps' evidence partials element-wise instead of the two doubles k_tail_data formed; no k_control launch).
    VBNMF_NO_CONTROL_FOLD=1 (read when an engine is created) restores the separate kernels: same steps, same stop, the same
    factors to rounding -- the evidence is summed over partitions first and over workgroups second, not the other way round."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from ccfindr_amd.parallel import cell_partition
    n, m, r, P = 260, 430, 7, 3
    X = _matrix(n, m, 91)
    wh = synth.random_state(n, m, r, HY, seed=8)
    M = C.CountMatrix(X)
    kw = dict(Itmax=itmax, Tol=tol, n0=10, dn=1, flags=(True,) * 4, history=True)
    out = {}
    for mode in ("fold", "separate"):
        if mode == "separate":
            monkeypatch.setenv("VBNMF_NO_CONTROL_FOLD", "1")
        else:
            monkeypatch.delenv("VBNMF_NO_CONTROL_FOLD", raising=False)
        comm, parts = _group(M, r, cell_partition(m, P), m, wh)
        res = comm.run(HY, **kw)
        again = comm.run(HY, **dict(kw, Itmax=3, Tol=0.0))            # a second run on the same engines starts from the state the first left
        out[mode] = (res, again, [p.get_state() for p in parts])
        for p in parts:
            p.close()
        comm.close()
    (a, a2, sa), (b, b2, sb) = out["fold"], out["separate"]
    assert a["it"] == b["it"] and a["reason"] == b["reason"] == (4 if tol == 0.0 else 2)
    assert a["it"] == itmax or tol > 0.0
    if tol > 0.0:
        assert 10 < a["it"] < itmax
    assert relerr(a["history"], b["history"]) <= 1e-11
    assert a2["it"] == b2["it"] == 3 and relerr(a2["history"], b2["history"]) <= 1e-11
    for x, y in zip(sa, sb):
        for k in x:
            assert relerr(x[k], y[k]) <= 1e-10, k
