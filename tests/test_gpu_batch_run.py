"""A batch of engines stepped by one launch (include/vbnmf.h: vbnmf_batch_run; csrc/kernels.h: k_update2_batch, k_sweep_batch).

The restarts of one rank on one matrix (reference R/bayesian.R:260-261, `lapply(seq_len(nrun), vb_iterate)`) share their
launches; every engine follows its own control block.  The bodies are the single engine's, so per engine:
  * iteration count, stop reason, lagging evidence, hyper-parameters, the whole history and the final state are those of
    VBEngine.run on that engine alone, BIT FOR BIT -- whatever the other engines of the batch do, however early they stop;
  * an engine that stops early idles through the others' steps and keeps the state its break left;
  * after the batch the engines serve host-stepped steps and further runs as before;
  * vb_factorize(batch=B) returns what the sequential driver returns.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HY = {"aw": 1.1, "bw": 0.9, "ah": 0.8, "bh": 1.3}


def _matrix(kind, n, m, seed):
    from ccfindr_amd import synth
    if kind == "clustered":
        return synth.fill_empty(synth.simulate_data(n, [m // 3, m - m // 3], alpha0=0.2, seed=seed, depth=np.full(m, 80)))
    rng = np.random.default_rng(seed)
    X = rng.poisson(0.6, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1.0
    X[rng.integers(0, n, m), np.arange(m)] += 1.0
    if kind == "noninteger":                                   # the wide layout (value + index streams)
        X = X * rng.uniform(0.5, 1.5, size=(1, m))
    return np.asfortranarray(X)


def _alone(M, r, wh, hy, grid=None, **kw):
    import ccfindr_amd as C
    eng = C.VBEngine(M, r, grid=grid)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    out = eng.run(hy, history=True, **kw)
    st = eng.get_state()
    eng.close()
    return out, st


def _same(a, b):
    assert a["it"] == b["it"] and a["reason"] == b["reason"]
    assert a["lk0"] == b["lk0"] or (np.isnan(a["lk0"]) and np.isnan(b["lk0"]))
    assert a["lkh"] == b["lkh"] or (np.isnan(a["lkh"]) and np.isnan(b["lkh"]))
    assert a["hyper"] == b["hyper"]
    assert np.array_equal(a["history"], b["history"], equal_nan=True)


@pytest.mark.parametrize("kind,n,m,r,B,own_grid", [("clustered", 400, 650, 5, 4, True), ("counts", 150, 230, 3, 7, True),
                                                    ("noninteger", 120, 260, 7, 3, False), ("counts", 97, 131, 16, 2, True),
                                                    ("clustered", 300, 500, 10, 5, False), ("counts", 64, 64, 2, 1, True),
                                                    ("clustered", 500, 700, 4, 32, True)])
def test_every_engine_of_a_batch_gives_its_stand_alone_run_bit_for_bit(kind, n, m, r, B, own_grid):
    """own_grid: the engines carry the grid a batch of B wants (256 / B workgroups and blocks, C.batch_grid) -- the stand-alone
    engines it is held to as well, the grid being part of the summation order; False: the default grids (one per CU)."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = _matrix(kind, n, m, 3 * r + n)
    n, m = X.shape
    M = C.CountMatrix(X)
    grid = C.batch_grid(B) if own_grid else None
    hys = [dict(HY, aw=HY["aw"] * (1 + 0.02 * b), bh=HY["bh"] * (1 - 0.01 * b)) for b in range(B)]     # (per-engine hyper-parameters)
    whs = [synth.random_state(n, m, r, hys[b], seed=10 + b) for b in range(B)]
    kw = dict(Itmax=120, Tol=2e-4, n0=4, dn=1)                  # the engines stop at different steps; some reach Itmax
    want = [_alone(M, r, whs[b], hys[b], grid=grid, **kw) for b in range(B)]
    engs = [C.VBEngine(M, r, grid=grid) for _ in range(B)]
    for eng, wh in zip(engs, whs):
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    got = C.run_batch(engs, hys, history=True, **kw)
    for b in range(B):
        _same(got[b], want[b][0])
        st = engs[b].get_state()
        for k in st:
            assert np.array_equal(st[k], want[b][1][k]), (b, k)
    # the engines go on: a host-stepped step and a second (stand-alone) run from the state the batch left
    ref = C.VBEngine(M, r, grid=grid)
    ref.set_state(whs[0]["lw"], whs[0]["lh"], whs[0]["eh"])
    ref.run(hys[0], **kw)
    assert engs[0].step(got[0]["hyper"])[0] == ref.step(want[0][0]["hyper"])[0]
    a = engs[0].run(got[0]["hyper"], Itmax=9, Tol=0.0, n0=2, history=True)
    b_ = ref.run(want[0][0]["hyper"], Itmax=9, Tol=0.0, n0=2, history=True)
    _same(a, b_)
    # ... and a second batch on the same engines
    again = C.run_batch(engs, [g["hyper"] for g in got], Itmax=6, Tol=0.0, n0=1, history=True)
    assert all(o["it"] == 6 and o["reason"] == 4 for o in again)
    ref.close()
    for eng in engs:
        eng.close()
    M.close()


@pytest.mark.parametrize("flags", [(True,) * 4, (False,) * 4, (True, False, False, True)])
def test_max_it_and_the_closing_control_step(flags):
    """Tol = 0: every engine runs to Itmax (reason 4) and the step behind the last one is evaluated by the closing launches."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = _matrix("clustered", 260, 330, 5)
    n, m = X.shape
    M = C.CountMatrix(X)
    r, B = 4, 3
    whs = [synth.random_state(n, m, r, HY, seed=b) for b in range(B)]
    for Itmax in (1, 2, 8, 17):                                 # inside the first batch of eight, at its edge, beyond
        kw = dict(Itmax=Itmax, Tol=0.0, n0=3, dn=2, flags=flags)
        want = [_alone(M, r, whs[b], HY, **kw) for b in range(B)]
        engs = [C.VBEngine(M, r) for _ in range(B)]
        for eng, wh in zip(engs, whs):
            eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        got = C.run_batch(engs, [HY] * B, history=True, **kw)
        for b in range(B):
            assert got[b]["it"] == Itmax and got[b]["reason"] == 4
            _same(got[b], want[b][0])
            st = engs[b].get_state()
            for k in st:
                assert np.array_equal(st[k], want[b][1][k]), (Itmax, b, k)
        for eng in engs:
            eng.close()
    M.close()


def test_a_nan_engine_stops_alone():
    """One engine of the batch starts from a state whose evidence is NaN (a whole factor row 0 with fudge = 0, reference
    src/vbnmf_update.cpp:34): it breaks with reason 1 at its first step (R/bayesian.R:345); the others are not touched."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = _matrix("counts", 200, 300, 5)
    n, m = X.shape
    M = C.CountMatrix(X)
    r, B = 3, 3
    whs = [synth.random_state(n, m, r, HY, seed=b) for b in range(B)]
    whs[1]["lw"][0, :] = 0.0
    kw = dict(Itmax=25, Tol=1e-5, fudge=0.0, flags=(False,) * 4)
    want = [_alone(M, r, whs[b], HY, **kw) for b in range(B)]
    assert want[1][0]["reason"] == 1
    engs = [C.VBEngine(M, r) for _ in range(B)]
    for eng, wh in zip(engs, whs):
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    got = C.run_batch(engs, [HY] * B, history=True, **kw)
    for b in range(B):
        _same(got[b], want[b][0])
    for eng in engs:
        eng.close()
    M.close()


def test_what_a_batch_refuses():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = _matrix("counts", 120, 160, 2)
    n, m = X.shape
    M = C.CountMatrix(X)
    a, b = C.VBEngine(M, 3), C.VBEngine(M, 5)
    wh3, wh4 = synth.random_state(n, m, 3, HY, seed=1), synth.random_state(n, m, 5, HY, seed=1)
    a.set_state(wh3["lw"], wh3["lh"], wh3["eh"])
    blank = C.VBEngine(M, 3)
    with pytest.raises(C.VBNMFError):                          # an engine without a state
        C.run_batch([a, blank], [HY, HY], Itmax=3)
    b.set_state(wh4["lw"], wh4["lh"], wh4["eh"])
    with pytest.raises(C.VBNMFError):                          # two row widths (ranks 3 and 5: four and six columns)
        C.run_batch([a, b], [HY, HY], Itmax=3)
    with pytest.raises(C.VBNMFError):                          # the same engine twice
        C.run_batch([a, a], [HY, HY], Itmax=3)
    big = C.VBEngine(M, 20)
    wh20 = synth.random_state(n, m, 20, HY, seed=1)
    big.set_state(wh20["lw"], wh20["lh"], wh20["eh"])
    with pytest.raises(C.VBNMFError):                          # beyond the batch kernels' ranks
        C.run_batch([big], [HY], Itmax=3)
    assert C.run_batch([a], [HY], Itmax=3, Tol=0.0)[0]["it"] == 3
    for e in (a, b, big, blank):
        e.close()
    M.close()


@pytest.mark.parametrize("unif_stop", [False, True])
def test_vb_factorize_batched_is_the_sequential_driver(unif_stop):
    import warnings
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.drop_empty(synth.simulate_data(400, (100, 150, 250), seed=9, sparse=True))
    M = C.CountMatrix(X)
    kw = dict(ranks=range(2, 7), nrun=5, verbose=0, Tol=1e-5, seed=11, Itmax=400, unif_stop=unif_stop)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = C.vb_factorize(M, batch=1, grid=C.batch_grid(3), **kw)       # one unit at a time, on the grids a batch of 3 uses
        b = C.vb_factorize(M, batch=3, across_ranks=False, **kw)          # the restarts of a rank, chunks of 3 + 2
        c = C.vb_factorize(M, **kw)                                       # the default at this size: 16 units at a time, across ranks
        nb = C.engine.auto_batch(M.nnz, 5 * 5)
        d = C.vb_factorize(M, batch=1, grid=C.batch_grid(nb), pad_rank=C.engine.padded_rank(6), **kw)
        e = C.vb_factorize(M, batch=1, **kw)                              # the default grids (one per CU), every rank its own width
    for one, other in ((a, b), (d, c)):
        assert one.ranks == other.ranks and one.measure == other.measure and one.nsteps == other.nsteps
        for x, y in zip(one.basis + one.coeff + one.dbasis + one.dcoeff, other.basis + other.coeff + other.dbasis + other.dcoeff):
            assert np.array_equal(x, y)
    # different grids: the block-wise sums are added in another order -- the same factorisation to rounding
    assert e.ranks == c.ranks
    assert np.allclose(e.measure["lml"], c.measure["lml"], rtol=1e-7, atol=0.0)
    M.close()


def test_the_grid_is_a_property_of_the_engines_the_thread_creates_next():
    import ccfindr_amd as C
    from ccfindr_amd import _native as N
    X = _matrix("counts", 120, 160, 2)
    M = C.CountMatrix(X)
    L = N.load()
    with pytest.raises(C.VBNMFError):
        N.check(L.vbnmf_set_engine_grid(-1, 0))
    with pytest.raises(C.VBNMFError):
        N.check(L.vbnmf_set_engine_grid(8, 100000))
    small = C.VBEngine(M, 3, grid=(16, 16))
    plain = C.VBEngine(M, 3)                                   # the hint is gone once the engine exists
    from ccfindr_amd import synth
    wh = synth.random_state(*X.shape, 3, HY, seed=1)
    out = []
    for eng in (small, plain):
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        out.append([eng.step(HY)[0] for _ in range(3)])
        eng.close()
    assert np.allclose(out[0], out[1], rtol=1e-12, atol=0.0)
    with pytest.raises(C.VBNMFError):                          # engines of different grids do not share a batch
        a, b = C.VBEngine(M, 3, grid=(16, 16)), C.VBEngine(M, 3)
        try:
            for eng in (a, b):
                eng.set_state(wh["lw"], wh["lh"], wh["eh"])
            C.run_batch([a, b], [HY, HY], Itmax=2)
        finally:
            a.close(); b.close()
    M.close()


# ---- maximum-likelihood NMF: the restarts of factorize() (reference R/factorize.R:181) stepped together ----------------
@pytest.mark.parametrize("kind,n,m,r,B,prior", [("clustered", 400, 650, 5, 4, False), ("counts", 150, 230, 3, 7, False),
                                                 ("noninteger", 120, 260, 7, 3, True), ("counts", 97, 131, 16, 2, False),
                                                 ("clustered", 300, 500, 10, 16, False), ("counts", 64, 64, 2, 1, False)])
def test_ml_batch_gives_every_engines_stand_alone_run_bit_for_bit(kind, n, m, r, B, prior):
    import ccfindr_amd as C
    X = _matrix(kind, n, m, 5 * r + n)
    n, m = X.shape
    M = C.CountMatrix(X)
    rng = np.random.default_rng(r)
    starts = [(rng.uniform(0.1, 1.0, size=(n, r)), rng.uniform(0.1, 1.0, size=(r, m))) for _ in range(B)]
    grid = C.batch_grid(B)
    for Itmax, Tol in ((60, 1e-4), (1, 0.0), (8, 0.0), (19, 0.0)):          # stops inside the run; Itmax inside / at / beyond a batch of eight
        kw = dict(Itmax=Itmax, Tol=Tol, prior=prior, gamma_a=1.3, gamma_b=0.7)
        want = []
        for w0, h0 in starts:
            eng = C.VBEngine(M, r, grid=grid)
            eng.ml_set_state(w0, h0)
            out = eng.ml_run(history=True, **kw)
            want.append((out, eng.ml_get_state(), eng.ml_likelihood()))
            eng.close()
        engs = [C.VBEngine(M, r, grid=grid) for _ in range(B)]
        for eng, (w0, h0) in zip(engs, starts):
            eng.ml_set_state(w0, h0)
        got = C.run_batch_ml(engs, history=True, **kw)
        for b in range(B):
            assert got[b]["it"] == want[b][0]["it"] and got[b]["reason"] == want[b][0]["reason"] and got[b]["lk"] == want[b][0]["lk"]
            assert np.array_equal(got[b]["history"], want[b][0]["history"])
            st = engs[b].ml_get_state()
            assert np.array_equal(st["ew"], want[b][1]["ew"]) and np.array_equal(st["eh"], want[b][1]["eh"])
            assert engs[b].ml_likelihood() == want[b][2]
        # the engines go on: a host-stepped ML step from the state the batch left, against the stand-alone engine's
        ref = C.VBEngine(M, r, grid=grid)
        ref.ml_set_state(*starts[0])
        ref.ml_run(**kw)
        assert engs[0].ml_step(prior, 1.3, 0.7) == ref.ml_step(prior, 1.3, 0.7)
        ref.close()
        for eng in engs:
            eng.close()
    M.close()


def test_factorize_batched_is_the_run_by_run_driver():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.drop_empty(synth.simulate_data(300, (80, 120, 100), seed=4, sparse=False))
    kw = dict(ranks=[2, 3, 4], nrun=7, verbose=0, Tol=1e-6, Itmax=400, seed=5)
    a = C.factorize(X, batch=3, **kw)                                  # chunks of 3 + 3 + 1
    M = C.CountMatrix(X)
    # one restart at a time on the grids a batch of three uses: the same draws, the same runs, the same best
    rng = np.random.default_rng(5)
    from ccfindr_amd.factorize import init
    for irank, rank in enumerate(kw["ranks"]):
        best, steps = (-np.inf, None), []
        for irun in range(kw["nrun"]):
            wh = init(X.shape[0], X.shape[1], rank, rng)
            eng = C.VBEngine(M, rank, grid=C.batch_grid(3))
            eng.ml_set_state(wh["ew"], wh["eh"])
            out = eng.ml_run(Itmax=400, Tol=1e-6)
            steps.append(out["it"])
            if irun == 0 or out["lk"] > best[0]:
                best = (out["lk"], eng.ml_get_state())
            eng.close()
        assert a.nsteps[irank] == steps
        assert a.measure["likelihood"][irank] == best[0]
        assert np.array_equal(a.basis[irank], best[1]["ew"]) and np.array_equal(a.coeff[irank], best[1]["eh"])
    M.close()
    b = C.factorize(X, batch=1, **kw)                                  # default grids: the same factorisation to rounding
    c = C.factorize(X, **kw)                                           # the default (batched at this size)
    assert np.allclose(a.measure["likelihood"], b.measure["likelihood"], rtol=1e-9, atol=0.0)
    assert np.allclose(c.measure["likelihood"], b.measure["likelihood"], rtol=1e-9, atol=0.0)
    with pytest.raises(ValueError):
        C.factorize(X, ranks=[2], nrun=1, batch=4, verbose=0, Itmax=5)


# ---- engines of several ranks in one batch: one row width (vbnmf_set_engine_padding) -----------------------------------------
def test_a_batch_spans_ranks_when_the_engines_are_one_width():
    """Engines of ranks 2..8 made eight columns wide share kernels, layouts and update table: one batch; per engine the stand-alone
    run on an engine of the same width, bit for bit; against the engine of the rank's own width, the same step to rounding."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = _matrix("clustered", 400, 650, 17)
    n, m = X.shape
    M = C.CountMatrix(X)
    ranks = [2, 3, 4, 5, 7, 8]
    pad = C.engine.padded_rank(max(ranks))
    assert pad == 8
    grid = C.batch_grid(len(ranks))
    whs = [synth.random_state(n, m, r, HY, seed=r) for r in ranks]
    kw = dict(Itmax=90, Tol=3e-4, n0=4, dn=1)
    want = []
    for r, wh in zip(ranks, whs):
        eng = C.VBEngine(M, r, grid=grid, pad_rank=pad, geometry_rank=max(ranks))
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        want.append((eng.run(HY, history=True, **kw), eng.get_state()))
        eng.close()
    engs = [C.VBEngine(M, r, grid=grid, pad_rank=pad, geometry_rank=max(ranks)) for r in ranks]
    for eng, wh in zip(engs, whs):
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    got = C.run_batch(engs, [HY] * len(ranks), history=True, **kw)
    for b, r in enumerate(ranks):
        _same(got[b], want[b][0])
        st = engs[b].get_state()
        assert st["lw"].shape == (n, r) and st["lh"].shape == (r, m)
        for k in st:
            assert np.array_equal(st[k], want[b][1][k]), (r, k)
    for eng in engs:
        eng.close()
    # a padded engine against the rank's own width: three steps agree to rounding
    for r, wh in zip(ranks[:3], whs[:3]):
        lks = []
        for pr in (pad, None):
            eng = C.VBEngine(M, r, pad_rank=pr)
            eng.set_state(wh["lw"], wh["lh"], wh["eh"])
            lks.append([eng.step(HY)[0] for _ in range(3)])
            eng.close()
        assert np.allclose(lks[0], lks[1], rtol=1e-12, atol=0.0)
    with pytest.raises(C.VBNMFError):
        C.VBEngine(M, 3, pad_rank=7)                           # not a padded rank
    M.close()


@pytest.mark.parametrize("nrun,unif_stop", [(1, True), (3, False)])
def test_vb_factorize_batched_across_ranks_is_the_unit_by_unit_driver(nrun, unif_stop):
    """nrun = 1 is the reference's default: the units of a small matrix's rank sweep are then its ranks.  The batched driver
    (units grouped over consecutive ranks) against one unit at a time on engines of the same grid and width."""
    import warnings
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.drop_empty(synth.simulate_data(400, (100, 150, 250), seed=9, sparse=True))
    M = C.CountMatrix(X)
    kw = dict(ranks=range(2, 9), nrun=nrun, verbose=0, Tol=1e-5, seed=11, Itmax=300, unif_stop=unif_stop)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = C.vb_factorize(M, **kw)                                                       # default: batched across ranks at this size
        nb = C.engine.auto_batch(M.nnz, nrun * 7)
        b = C.vb_factorize(M, batch=1, grid=C.batch_grid(nb), pad_rank=C.engine.padded_rank(8), **kw)
        c = C.vb_factorize(M, batch=1, **kw)                                              # the ranks' own widths, default grids
    assert a.ranks == b.ranks and a.measure == b.measure and a.nsteps == b.nsteps
    for x, y in zip(a.basis + a.coeff, b.basis + b.coeff):
        assert np.array_equal(x, y)
    assert a.ranks == c.ranks and np.allclose(a.measure["lml"], c.measure["lml"], rtol=1e-7, atol=0.0)
    M.close()
