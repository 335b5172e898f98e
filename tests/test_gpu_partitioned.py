"""The cell-partitioned engine path on ONE GPU: two partitions in one process, their reduce
buffers summed by hand where RCCL's all-reduce would go; must equal the unpartitioned engine."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HY = {"aw": 1.2, "bw": 0.9, "ah": 0.8, "bh": 1.5}


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.mark.parametrize("n,m,r,cut", [(120, 260, 4, 130), (300, 501, 10, 200), (64, 90, 1, 17)])
def test_two_partitions_equal_whole(n, m, r, cut):
    import torch
    import ccfindr_amd as C
    from ccfindr_amd import synth
    rng = np.random.default_rng(n + m)
    X = rng.poisson(0.3, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1
    X[rng.integers(0, n, m), np.arange(m)] += 1
    wh = synth.random_state(n, m, r, HY, seed=3)
    M = C.CountMatrix(X)
    whole = C.VBEngine(M, r)
    whole.set_state(wh["lw"], wh["lh"], wh["eh"])
    parts = [C.VBEngine(M, r, cols=(0, cut), m_global=m), C.VBEngine(M, r, cols=(cut, m), m_global=m)]
    reds = [p.reduce_tensor() for p in parts]
    assert reds[0].numel() == n * ((r + 1) // 2 * 2) + ((r + 1) // 2 * 2) + 4

    def allreduce():
        torch.cuda.synchronize()
        s = reds[0] + reds[1]
        reds[0].copy_(s); reds[1].copy_(s)
        torch.cuda.synchronize()

    for p, (b, e) in zip(parts, ((0, cut), (cut, m))):
        p.set_state(wh["lw"], wh["lh"][:, b:e], wh["eh"][:, b:e])
    allreduce()
    for p in parts:
        p.state_finish()
    for _ in range(6):
        lkh0, st0 = whole.step(HY)
        for p in parts:
            p.step_local(HY)
        allreduce()
        outs = [p.step_finish() for p in parts]
        assert outs[0] == outs[1]                                   # replicated scalars are bit-identical
        assert abs(outs[0][0] / lkh0 - 1) <= 1e-11
        assert np.allclose(outs[0][1], st0, rtol=1e-11)
    ref = whole.get_state()
    a, b = parts[0].get_state(), parts[1].get_state()
    for k in ("lw", "ew", "dw"):
        assert np.array_equal(a[k], b[k])                           # gene side replicated
        assert relerr(a[k], ref[k]) <= 1e-11
    for k in ("lh", "eh", "dh"):
        assert relerr(np.concatenate([a[k], b[k]], axis=1), ref[k]) <= 1e-11
    with pytest.raises(C.VBNMFError):
        parts[0].step(HY)                                           # a partitioned engine refuses the unsplit step
    for e in parts + [whole]:
        e.close()


def test_cell_partitioned_engine_over_rccl_world1():
    """CellPartitionedEngine with a real (1-rank) RCCL group, both exchange paths: torch.distributed's all-reduce on the
    engine's HIP stream through ExternalStream (native=False), and the library's own ncclAllReduce (native=True)."""
    import os
    import torch
    import torch.distributed as dist
    import ccfindr_amd as C
    from ccfindr_amd import parallel, synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(5)
        X = rng.poisson(0.4, size=(90, 140)).astype(np.float64) + np.eye(90, 140)
        X[0, :] += 1
        M = C.CountMatrix(X)
        wh = synth.random_state(90, 140, 3, HY, seed=4)
        eng = parallel.CellPartitionedEngine(M, 3, device=0, native=False)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        dflt = parallel.CellPartitionedEngine(M, 3, device=0)         # one process: no communicator by default
        assert not dflt.native and dflt.comm is None
        dflt.close()
        nat = parallel.CellPartitionedEngine(M, 3, device=0, native=True)
        assert nat.native and nat.comm.kind == "rccl"
        nat.set_state(wh["lw"], wh["lh"], wh["eh"])
        ref = C.VBEngine(M, 3)
        ref.set_state(wh["lw"], wh["lh"], wh["eh"])
        for _ in range(4):
            # force the collective path even at world 1 (sum over one rank = identity)
            eng.engine.step_local(HY)
            with eng.engine.stream_context():
                dist.all_reduce(eng._red)
            got = eng.engine.step_finish()
            want = ref.step(HY)
            assert got == want
            assert nat.step(HY) == want                               # step_local / vbnmf_engine_allreduce / step_finish
        eng.close(); nat.close(); ref.close()
    finally:
        dist.destroy_process_group()
