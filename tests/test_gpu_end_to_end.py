"""GPU: the README's flow end to end -- 10x files -> native reader -> vb_factorize / factorize -> cluster ids -- on
simulated counts with three planted clusters (the reference's own usage pattern: read_10x, vb_factorize(ranks, nrun),
cluster_id; R/utils.R:28-54, R/bayesian.R:229-301, R/utils.R:903-909)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_files_to_clusters(tmp_path):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.drop_empty(synth.simulate_data(300, (60, 90, 150), seed=4, sparse=True))
    n, m = X.shape
    genes = [[f"ENSG{i:011d}", f"G{i}"] for i in range(n)]
    cells = [[f"CELL{j}-1"] for j in range(m)]
    C.write_10x(C.CountData(X.tocsc(), genes, cells), str(tmp_path))
    d = C.read_10x(str(tmp_path))
    assert d.counts.shape == (n, m) and (d.counts != X.tocsc()).nnz == 0
    M = d.count_matrix()
    vb = C.vb_factorize(M, ranks=range(2, 6), nrun=2, verbose=0, Tol=1e-5, seed=1, Itmax=400)
    assert vb.ranks == [2, 3, 4, 5] and len(vb.basis) == 4
    lml = vb.measure["lml"]
    assert int(np.argmax(lml)) >= 1                          # the evidence does not pick too few components
    cid = C.cluster_id(vb, rank=3)
    assert cid.shape == (m,) and set(cid.tolist()) <= {1, 2, 3}
    # the planted clusters are contiguous column ranges in simulate_data's output before shuffling; whatever the order,
    # three planted groups must come back as three dominant labels
    assert len(set(cid.tolist())) == 3
    ml = C.factorize(d.counts, ranks=[2, 3], nrun=3, verbose=0, seed=2, Itmax=300, Tol=1e-5)
    assert len(ml.coeff) == 2 and ml.coeff[1].shape == (3, m)
    assert ml.measure["likelihood"][1] >= ml.measure["likelihood"][0]     # a larger rank never fits worse
    assert 0 < ml.measure["dispersion"][0] <= 1 and -1 <= ml.measure["cophenetic"][0] <= 1
    M.close()


@pytest.mark.parametrize("host_threads", [0, 1])
def test_concurrent_units_give_the_same_result(host_threads):
    """vb_factorize(concurrent=4): (run, rank) units overlap on the one GPU; every unit has its own seeded stream, so the
    outcome is that of the sequential driver, bit for bit.  With ONE library host thread per call (no threads started inside
    engine creation and set_state: the units' first kernels follow their engines' creation most closely) this is the
    configuration that exposed the null-stream fills of engine creation in round 5 (profiles/r05_concurrent_race.txt)."""
    import time
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from ccfindr_amd.engine import set_host_threads
    X = synth.drop_empty(synth.simulate_data(400, (100, 150, 250), seed=9, sparse=True))
    M = C.CountMatrix(X)
    kw = dict(ranks=range(2, 8), nrun=3, verbose=0, Tol=1e-6, seed=11, Itmax=1500, unif_stop=False)
    set_host_threads(host_threads)
    try:
        t0 = time.perf_counter(); a = C.vb_factorize(M, batch=1, **kw); ta = time.perf_counter() - t0      # (one unit at a time, the default grids)
        t0 = time.perf_counter(); b = C.vb_factorize(M, concurrent=4, **kw); tb = time.perf_counter() - t0
    finally:
        set_host_threads(0)
    assert a.ranks == b.ranks and a.measure == b.measure and a.nsteps == b.nsteps
    for x, y in zip(a.basis + a.coeff, b.basis + b.coeff):
        assert np.array_equal(x, y)
    print(f"sequential {ta:.2f} s, concurrent=4 {tb:.2f} s")
    M.close()


def test_an_engine_created_while_the_null_stream_is_busy_starts_from_the_same_buffers():
    """Engine creation zeroes its block partials, column sums and `ew / dw / dh`.  Until round 5 it did so with hipMemset, which
    on this ROCm returns at once and runs on the device's NULL stream -- which the engine's non-blocking stream does not wait
    for: with the null stream backed up (other host threads creating engines, vb_factorize(concurrent=K); here: a second
    host thread queueing large fills through torch, whose default stream is the null stream) the zeros landed AFTER the priming kernels of
    set_state had written the same buffers, and the first step started from zeroed partials.  The fills are on the engine's own
    stream now: an engine created under that load must give the bits of one created on an idle device.  (The invariant, not a
    reproducer: the old library passes this too on most boxes -- the failure needed several engine-creating host threads,
    test_concurrent_units_give_the_same_result under VBNMF_HOST_THREADS=1, profiles/r05_concurrent_race.txt.)"""
    import torch
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.drop_empty(synth.simulate_data(300, (120, 200), seed=4, sparse=True))
    n, m = X.shape
    r = 4
    hy = {"aw": 0.9, "bw": 1.1, "ah": 1.2, "bh": 0.8}
    wh = synth.random_state(n, m, r, hy, seed=5)
    M = C.CountMatrix(X)

    def three_steps():
        eng = C.VBEngine(M, r)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        lk = [eng.step(hy)[0] for _ in range(3)]
        st = eng.get_state()
        eng.close()
        return lk, st

    import threading
    quiet = three_steps()
    big = torch.empty(2 << 30, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    stop = threading.Event()

    def keep_the_null_stream_busy():                             # a queue of fills, two batches (~7 ms) deep, until told to stop
        pending = []
        while not stop.is_set():
            for _ in range(10):
                big.zero_()
            ev = torch.cuda.Event()
            ev.record()
            pending.append(ev)
            if len(pending) > 2:
                pending.pop(0).synchronize()
        torch.cuda.synchronize()

    th = threading.Thread(target=keep_the_null_stream_busy)
    th.start()
    try:
        busy = [three_steps() for _ in range(3)]
    finally:
        stop.set()
        th.join()
    del big
    for lk, st in busy:
        assert lk == quiet[0]
        for k in quiet[1]:
            assert np.array_equal(st[k], quiet[1][k]), k
    M.close()


def test_layouts_uploaded_ahead_of_the_first_engine_change_nothing():
    """vbnmf_device_warmup / vbnmf_matrix_prepare_async / vbnmf_matrix_preload_layout only move work earlier (the sharded
    sweep's peers spend their wait for the layouts there): an engine created afterwards is bit-identical to one created cold,
    and a second preload of the same geometry is a no-op."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from ccfindr_amd.engine import device_warmup, sweep_workgroups
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    X = synth.fill_empty(synth.simulate_data(500, [300, 400], alpha0=0.2, seed=21, depth=np.full(700, 100)), seed=21)
    n, m = X.shape
    wh = synth.random_state(n, m, 6, hy, seed=4)
    outs = []
    for warm in (False, True):
        M = C.CountMatrix(X)
        if warm:
            device_warmup(0)
            M.prepare_async()
            n_wg = sweep_workgroups(0)
            for side in (1, 0):
                M.preload_layout(side, 8, n_wg, 0)
                M.preload_layout(side, 8, n_wg, 0)
            M.prepare()
        eng = C.VBEngine(M, 6, geometry_rank=8)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        res = eng.run(hy, Itmax=12, Tol=0.0, flags=(True,) * 4, history=True)
        outs.append((res["history"], eng.get_state()))
        eng.close(); M.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k


def test_device_buffers_are_pooled_between_engines_and_trimmed_on_request():
    """An engine's device buffers go to the library's pool when it is destroyed (a rank sweep creates one engine per unit and
    every hipFree synchronises the device) and vbnmf_pool_trim returns them to the driver; results do not depend on whether
    an engine's buffers are fresh or reused (the arrays the kernels read are written or cleared first)."""
    import torch
    import ccfindr_amd as C
    from ccfindr_amd import synth
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    X = synth.fill_empty(synth.simulate_data(800, [600, 700], alpha0=0.2, seed=23, depth=np.full(1300, 120)), seed=23)
    n, m = X.shape
    M = C.CountMatrix(X)
    wh = {r: synth.random_state(n, m, r, hy, seed=r) for r in (12, 7)}
    C.load().vbnmf_pool_trim()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]

    def run(r):
        eng = C.VBEngine(M, r)
        eng.set_state(wh[r]["lw"], wh[r]["lh"], wh[r]["eh"])
        out = eng.run(hy, Itmax=9, Tol=0.0, flags=(True,) * 4, history=True)
        st = eng.get_state()
        eng.close()
        return out["history"], st

    first = {r: run(r) for r in (12, 7)}                       # rank 7 reuses (larger) buffers rank 12 left behind
    again = {r: run(r) for r in (7, 12)}                       # ... and now the other way round
    for r in (12, 7):
        assert np.array_equal(first[r][0], again[r][0])
        for k in first[r][1]:
            assert np.array_equal(first[r][1][k], again[r][1][k]), (r, k)
    M.close()                                                  # (what waits in the pool is below the driver's mapping granularity here)
    C.load().vbnmf_pool_trim()
    torch.cuda.synchronize()
    assert abs(free0 - torch.cuda.mem_get_info()[0]) < 64 * 2**20
