"""N > 1 paths over gloo on the CPU (world_size 2): the cell-partition protocol and the sharded
rank sweep, with the numpy engine standing in for the HIP engine."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HY = {"aw": 1.2, "bw": 0.9, "ah": 0.8, "bh": 1.5}


def _data():
    from ccfindr_amd import synth
    return synth.drop_empty(synth.simulate_data(50, (30, 45), seed=6, sparse=False))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_engine import NumpyPhaseEngine
        from ccfindr_amd import parallel, synth
        X = _data()
        n, m = X.shape
        r = 3
        wh = synth.random_state(n, m, r, HY, seed=9)
        # --- cell-partitioned factorisation
        cols = parallel.cell_partition(m, world)[rank]
        eng = parallel.CellPartitionedEngine(X, r, engine=NumpyPhaseEngine(X, r, cols=cols, m_global=m))
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        trace = [eng.step(HY) for _ in range(6)]
        state = eng.get_state()
        # --- sharded rank sweep
        res = parallel.vb_factorize_sharded(X, ranks=[2, 3, 4], nrun=2, Itmax=25, seed=11,
                                            engine_factory=lambda M, rk: NumpyPhaseEngine(X, rk))
        q.put((rank, trace, state, (res.ranks, res.measure, res.nsteps, [b.copy() for b in res.basis])))
    finally:
        dist.destroy_process_group()


def test_cell_partition_and_sharded_sweep_world2():
    sys.path.insert(0, HERE)
    from fake_engine import NumpyPhaseEngine
    import ccfindr_amd as C
    from ccfindr_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(k, 2, port, q)) for k in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    X = _data()
    n, m = X.shape
    wh = synth.random_state(n, m, 3, HY, seed=9)
    ref = NumpyPhaseEngine(X, 3)
    ref.set_state(wh["lw"], wh["lh"], wh["eh"])
    ref_trace = [ref.step(HY) for _ in range(6)]
    ref_state = ref.get_state()
    for rank, trace, state, _ in outs:
        for (lkh, st), (lkh0, st0) in zip(trace, ref_trace):
            assert lkh == pytest.approx(lkh0, rel=1e-12)
            assert np.allclose(st, st0, rtol=1e-12)
        for k in ref_state:                                   # lh/eh/dh are all-gathered to full width
            assert state[k].shape == ref_state[k].shape
            assert np.allclose(state[k], ref_state[k], rtol=1e-11), k
    assert outs[0][1] == outs[1][1]                           # identical on every partition

    serial = C.vb_factorize(X, ranks=[2, 3, 4], nrun=2, verbose=0, Itmax=25, seed=11,
                            engine_factory=lambda M, rk: NumpyPhaseEngine(X, rk))
    for rank, _, _, (ranks, measure, nsteps, basis) in outs:
        assert ranks == serial.ranks and nsteps == serial.nsteps
        assert measure["lml"] == serial.measure["lml"]        # same units, same seeds: bit-identical
        for a, b in zip(basis, serial.basis):
            assert np.array_equal(a, b)


def _failing_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_engine import NumpyPhaseEngine
        from ccfindr_amd import parallel
        X = _data()

        class Failing(NumpyPhaseEngine):
            def step(self, hyper, fudge=None):
                raise RuntimeError("Hyper-parameter update failed to converge")

        # LPT order over ranks [4, 3, 2] on two processes: process 0 takes rank 4, process 1 ranks 3 and 2;
        # the stand-in engine of rank 3 fails, so only process 1 sees the error directly
        factory = lambda M, rk: (Failing if rk == 3 else NumpyPhaseEngine)(X, rk)
        try:
            parallel.vb_factorize_sharded(X, ranks=[2, 3, 4], nrun=1, Itmax=5, seed=11, engine_factory=factory)
            q.put((rank, "no error"))
        except parallel.ShardedRunError as exc:
            q.put((rank, str(exc)))
    finally:
        dist.destroy_process_group()


def test_sharded_sweep_unit_failure_reaches_every_rank():
    """A unit that raises on ONE process must not leave the others waiting in the gather: every process raises."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_failing_worker, args=(k, 2, port, q)) for k in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[0][1] == outs[1][1]
    assert "rank 3" in outs[0][1] and "failed to converge" in outs[0][1] and "process 1" in outs[0][1]


def _two_node_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    if os.environ.get("TEST_TINY_SHM"):
        os.environ["VBNMF_TEST_SHM_FREE"] = "4096"                  # one node, but its /dev/shm is (said to be) full
    else:
        os.environ["VBNMF_NODE_KEY"] = f"pretend-node-{rank}"      # no shared /dev/shm between the two processes
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_engine import NumpyPhaseEngine
        from ccfindr_amd import parallel
        X = _data()
        tm = {}
        res = parallel.vb_factorize_sharded(X, ranks=[2, 3, 4], nrun=1, Itmax=15, seed=11, timings=tm,
                                            engine_factory=lambda M, rk: NumpyPhaseEngine(X, rk))
        q.put((rank, res.ranks, res.measure, res.nsteps, [np.asarray(b).copy() for b in res.basis],
               [np.asarray(b).copy() for b in res.dcoeff], tm))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tiny_shm", [False, True])
def test_sharded_sweep_across_two_pretend_nodes_uses_tensor_broadcasts(tiny_shm, monkeypatch):
    """Processes that share no /dev/shm (forced through VBNMF_NODE_KEY) -- or whose /dev/shm has no room for the layouts and
    results (a container's 64 MB default; writing past a full tmpfs would be a SIGBUS), in which case the node is taken apart
    and every process works alone: a unit's factor matrices reach the others by tensor broadcast from its owner; the
    result is still the serial one, bit for bit, on every process."""
    if tiny_shm:
        monkeypatch.setenv("TEST_TINY_SHM", "1")
    sys.path.insert(0, HERE)
    from fake_engine import NumpyPhaseEngine
    import ccfindr_amd as C
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_two_node_worker, args=(k, 2, port, q)) for k in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X = _data()
    serial = C.vb_factorize(X, ranks=[2, 3, 4], nrun=1, verbose=0, Itmax=15, seed=11,
                            engine_factory=lambda M, rk: NumpyPhaseEngine(X, rk))
    for rank, ranks, measure, nsteps, basis, dcoeff, tm in outs:
        assert tm["node_processes"] == 1
        assert ranks == serial.ranks and nsteps == serial.nsteps and measure == serial.measure
        for a, b in zip(basis, serial.basis):
            assert np.array_equal(a, b)
        for a, b in zip(dcoeff, serial.dcoeff):
            assert np.array_equal(a, b)


def _uneven_worker(rank, world, port, q, mode):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    # nodes {0, 1} + {2}: processes 0 and 1 share their (pretend) node's memory file system, process 2 sits alone
    os.environ["VBNMF_NODE_KEY"] = "pretend-node-a" if rank < 2 else "pretend-node-b"
    if mode == "tiny_shm_on_a" and rank < 2:
        os.environ["VBNMF_TEST_SHM_FREE"] = "4096"                  # ... and node a's /dev/shm is (said to be) full: it is taken apart
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_engine import NumpyPhaseEngine
        from ccfindr_amd import parallel
        X = _data()
        tm = {}
        res = parallel.vb_factorize_sharded(X, ranks=[2, 3, 4, 5], nrun=1, Itmax=12, seed=11, timings=tm,
                                            engine_factory=lambda M, rk: NumpyPhaseEngine(X, rk))
        q.put((rank, res.ranks, res.measure, res.nsteps, [np.asarray(b).copy() for b in res.basis],
               [np.asarray(b).copy() for b in res.dcoeff], tm))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["uneven", "tiny_shm_on_a"])
def test_sharded_sweep_on_uneven_nodes_every_process_reaches_every_collective(mode):
    """World 3 with nodes {0, 1} + {2}: the result-segment setup used to call all_gather_object and barrier
    only on processes whose node has more than one member, so process 2 went on to the records' all-reduce while 0 and 1
    sat in the gather.  Also with node a short of /dev/shm only (it alone is taken apart).  The result is the serial one,
    bit for bit, on every process."""
    sys.path.insert(0, HERE)
    from fake_engine import NumpyPhaseEngine
    import ccfindr_amd as C
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33800 + (os.getpid() % 2000) + (7 if mode == "uneven" else 0)
    procs = [ctx.Process(target=_uneven_worker, args=(k, 3, port, q, mode)) for k in range(3)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X = _data()
    serial = C.vb_factorize(X, ranks=[2, 3, 4, 5], nrun=1, verbose=0, Itmax=12, seed=11,
                            engine_factory=lambda M, rk: NumpyPhaseEngine(X, rk))
    for rank, ranks, measure, nsteps, basis, dcoeff, tm in outs:
        assert tm["node_processes"] == (1 if (mode == "tiny_shm_on_a" or rank == 2) else 2)
        assert ranks == serial.ranks and nsteps == serial.nsteps and measure == serial.measure
        for a, b in zip(basis, serial.basis):
            assert np.array_equal(a, b)
        for a, b in zip(dcoeff, serial.dcoeff):
            assert np.array_equal(a, b)
