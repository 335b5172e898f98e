"""The library's RCCL protocol with MORE THAN ONE rank (missing #1): vbnmf_comm_create with nranks = 2,
queue_vb_step's RCCL branch (two out-of-place all-reduces per step, the n x R one travelling beside the cell-side sweep),
the event ring, drive_loop's replicated queueing, a convergence break inside a queued batch, the host-stepped
vbnmf_engine_allreduce, and the bounded waits when the peer is gone.

ONE test GPU and real RCCL refuses two ranks on a device, so the two processes open tests/fake_rccl's stand-in for librccl
(VBNMF_RCCL_LIB): same eight symbols, stream-ordered collectives through shared memory.  Everything above it -- the
communicator, the engines, the kernels, the queueing -- is the product's own code, untouched.  torch.distributed (gloo)
only carries the 128-byte id, as in a real run.  Reference analogue of the parallel driver: R/bayesian.R:262-263."""
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE = os.path.join(HERE, "fake_rccl", "_build", "libfake_rccl.so")
HY = {"aw": 1.2, "bw": 0.9, "ah": 0.8, "bh": 1.5}


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def _problem(kind):
    from ccfindr_amd import synth
    if kind == "converge":
        X = synth.drop_empty(synth.simulate_data(150, (120, 130, 150), seed=8, sparse=True))
        hy, r, kw = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}, 3, dict(Itmax=2000, Tol=1e-5, n0=10, dn=1, flags=(True,) * 4, history=True)
    else:
        X = synth.fill_empty(synth.simulate_data(600, [300, 350, 251], alpha0=0.2, seed=12, depth=np.full(901, 150)), seed=12)
        hy, r, kw = dict(HY), 6, dict(Itmax=37, Tol=0.0, n0=10, dn=1, flags=(True,) * 4, history=True)
    n, m = X.shape
    return X, n, m, r, hy, kw, synth.random_state(n, m, r, hy, seed=5)


def _worker(rank, world, port, kind, carrier, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VBNMF_RCCL_LIB"] = FAKE
    os.environ["FAKE_RCCL_TIMEOUT_S"] = "20"
    # carrier "kernel": the stand-in's collectives are KERNELS over IPC-mapped peer buffers, RCCL's launch shape (tens of
    # blocks x 512 threads, a few KB of LDS, spinning on the peers' flags); "host": a host function on the stream
    os.environ["FAKE_RCCL_KERNEL"] = "1" if carrier == "kernel" else "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ccfindr_amd as C
        from ccfindr_amd.parallel import CellPartitionedEngine
        X, n, m, r, hy, kw, wh = _problem(kind)
        M = C.CountMatrix(X)
        eng = CellPartitionedEngine(M, r, device=0, native=True)          # the library's communicator, 2 ranks
        assert eng.comm is not None and eng.comm.nranks == world
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])                        # state exchange: vbnmf_engine_allreduce
        out = eng.run(hy, **kw)                                            # device-driven loop, collectives queued from C++
        local = eng.engine.get_state()
        # the host-stepped protocol goes on from the loop's state
        lkh_next, stats_next = eng.step(out["hyper"])
        full = eng.get_state(("lh",))                                      # tensor all_gather of the cell blocks
        q.put((rank, eng.cols, {k: out[k] for k in ("it", "reason", "lk0", "lkh", "hyper", "history")},
               {k: local[k] for k in ("lw", "ew", "dw", "eh")}, lkh_next, stats_next, full["lh"].shape))
        eng.close()
    finally:
        dist.destroy_process_group()


def _spawn(target, args, world=2, timeout=300):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33400 + (os.getpid() % 1500)
    procs = [ctx.Process(target=target, args=(k, world, port) + args + (q,)) for k in range(world)]
    for p in procs:
        p.start()
    return procs, q


@pytest.mark.parametrize("carrier", ["host", "kernel"])
@pytest.mark.parametrize("kind", ["itmax", "converge"])
def test_two_ranks_of_the_rccl_protocol_equal_the_single_engine(kind, carrier):
    import ccfindr_amd as C
    assert os.path.exists(FAKE), "tests/fake_rccl is not built (make, or __graft_entry__.build())"
    procs, q = _spawn(_worker, (kind, carrier))
    outs = sorted([q.get(timeout=300) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, n, m, r, hy, kw, wh = _problem(kind)
    M = C.CountMatrix(X)
    whole = C.VBEngine(M, r)
    whole.set_state(wh["lw"], wh["lh"], wh["eh"])
    want = whole.run(hy, **kw)
    ref = whole.get_state()
    lkh1, stats1 = whole.step(want["hyper"])
    if kind == "converge":
        assert want["reason"] == 2 and 12 < want["it"] < 2000 and want["it"] % 8 != 0      # a break INSIDE a queued batch
    a, b = outs[0][2], outs[1][2]
    assert a["it"] == b["it"] == want["it"] and a["reason"] == b["reason"] == want["reason"]     # same stop on both ranks
    assert np.array_equal(a["history"], b["history"]) and a["hyper"] == b["hyper"] and a["lk0"] == b["lk0"]
    assert relerr(a["history"], want["history"]) <= 1e-10
    assert abs(a["lkh"] / want["lkh"] - 1) <= 1e-10 and abs(a["lk0"] / want["lk0"] - 1) <= 1e-10
    for k in ("lw", "ew", "dw"):
        assert np.array_equal(outs[0][3][k], outs[1][3][k]), k             # gene-side state bit-identical across ranks
        assert relerr(outs[0][3][k], ref[k]) <= 1e-9, k
    eh = np.concatenate([o[3]["eh"] for o in outs], axis=1)
    assert [o[1] for o in outs] == [(0, m // 2), (m // 2, m)] and relerr(eh, ref["eh"]) <= 1e-9
    for o in outs:
        assert abs(o[4] / lkh1 - 1) <= 1e-10 and relerr(np.asarray(o[5]), np.asarray(stats1)) <= 1e-10
        assert o[6] == (r, m)
    whole.close(); M.close()


def _dead_peer_worker(rank, world, port, when, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VBNMF_RCCL_LIB"] = FAKE
    os.environ["FAKE_RCCL_TIMEOUT_S"] = "12"            # the stand-in's own patience: longer than the library's bound
    os.environ["VBNMF_WAIT_TIMEOUT_S"] = "3"
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ccfindr_amd as C
    from ccfindr_amd import _native as N
    from ccfindr_amd.parallel import CellPartitionedEngine
    X, n, m, r, hy, kw, wh = _problem("itmax")
    M = C.CountMatrix(X)
    eng = CellPartitionedEngine(M, r, device=0, native=True)
    if when == "before_state" and rank == 1:
        os._exit(0)                                      # gone before the state exchange
    t0 = time.perf_counter()
    try:
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        if rank == 1:
            os._exit(0)                                  # gone before the loop: rank 0's first collective never completes
        eng.run(hy, **kw)
        q.put((rank, "no error", 0.0)); code = 0
    except N.VBNMFError as exc:
        q.put((rank, str(exc), time.perf_counter() - t0)); code = 3
    q.close(); q.join_thread()                           # the queue's feeder thread must have sent the record before ...
    os._exit(code)                                       # ... the process leaves without teardown (work may still be queued behind the dead collective)


@pytest.mark.parametrize("when", ["before_loop", "before_state"])
def test_a_dead_peer_ends_in_an_error_within_the_wait_bound(when):
    """Rank 1 exits early; rank 0's stream then sits behind a collective that never completes.  The library's bounded waits
    (VBNMF_WAIT_TIMEOUT_S = 3 s here) must turn that into VBNMF_ERR_HIP and a non-zero exit, not a hang."""
    assert os.path.exists(FAKE)
    procs, q = _spawn(_dead_peer_worker, (when,))
    rank, msg, waited = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
    assert rank == 0 and "timed out" in msg and "VBNMF_WAIT_TIMEOUT_S" in msg, msg
    assert 2.5 < waited < 9.0, waited
    assert procs[0].exitcode == 3 and procs[1].exitcode == 0
