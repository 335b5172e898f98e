"""GPU: the multi-process cell-partitioned path with the REAL engines -- two processes, each owning half of the cells,
both on the one GPU of the test box (RCCL refuses two ranks on one device, so the reduce buffer is all-reduced by gloo
through the host; everything else -- partition engines, step_local / step_finish, the sharded sweep -- is what the 8-GPU
run executes).  Results must equal the single-engine run."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HY = {"aw": 1.2, "bw": 0.9, "ah": 0.8, "bh": 1.5}


def _data():
    from ccfindr_amd import synth
    return synth.drop_empty(synth.simulate_data(300, (200, 300, 401), seed=6, sparse=True))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ccfindr_amd as C
        from ccfindr_amd import parallel, synth
        X = _data()
        n, m = X.shape
        r = 6
        wh = synth.random_state(n, m, r, HY, seed=9)
        eng = parallel.CellPartitionedEngine(C.CountMatrix(X), r, device=0)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        trace = [eng.step(HY) for _ in range(8)]
        state = eng.get_state()
        eng.close()
        res = parallel.vb_factorize_sharded(X, ranks=[2, 3, 4, 5], nrun=2, Itmax=60, seed=11, device=0)
        q.put((rank, trace, state, (res.ranks, res.measure, res.nsteps, [b.copy() for b in res.basis])))
    finally:
        dist.destroy_process_group()


def test_two_processes_one_gpu_equal_the_single_engine():
    import torch.multiprocessing as mp
    import ccfindr_amd as C
    from ccfindr_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(k, 2, port, q)) for k in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X = _data()
    n, m = X.shape
    wh = synth.random_state(n, m, 6, HY, seed=9)
    whole = C.VBEngine(C.CountMatrix(X), 6)
    whole.set_state(wh["lw"], wh["lh"], wh["eh"])
    ref_trace = [whole.step(HY) for _ in range(8)]
    ref = whole.get_state()
    whole.close()
    (_, t0, s0, f0), (_, t1, s1, f1) = outs
    assert t0 == t1                                                   # replicated scalars identical on both ranks
    for (lkh, st), (lkh_r, st_r) in zip(t0, ref_trace):
        assert abs(lkh / lkh_r - 1) <= 1e-11 and np.allclose(st, st_r, rtol=1e-11)
    for k in ("lw", "ew", "dw", "lh", "eh", "dh"):
        assert np.array_equal(s0[k], s1[k])
        assert np.max(np.abs(s0[k] - ref[k]) / np.abs(ref[k])) <= 1e-11, k
    # (batch=1: one unit at a time on the default grids, as the sharded driver's processes run them -- a batch's engines sit on
    # smaller grids, which fixes another order of the block-wise sums: the same results to rounding, not bit for bit)
    single = C.vb_factorize(X, ranks=[2, 3, 4, 5], nrun=2, Itmax=60, seed=11, verbose=0, batch=1)
    for f in (f0, f1):
        assert f[0] == single.ranks and f[2] == single.nsteps and f[1] == single.measure
        for a, b in zip(f[3], single.basis):
            assert np.array_equal(a, b)
