"""The product's host loop (ccfindr_amd.bayesian / parallel) on the CPU, with a numpy engine
standing in for the HIP engine (tests/fake_engine.py), checked against the oracle's restatement
of the reference driver (R/bayesian.R:2-53, 229-390)."""
import math
import warnings

import numpy as np
import pytest

from fake_engine import NumpyPhaseEngine
from oracle import vbnmf_oracle as O

HY1 = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}


def data(seed=4, n=60, groups=(40, 50)):
    from ccfindr_amd import synth
    return synth.drop_empty(synth.simulate_data(n, groups, seed=seed, sparse=False))


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def test_numpy_engine_matches_literal_oracle():
    """The fused-step algebra (carried statistics, collapsed sums, evidence identity) == the literal step."""
    from ccfindr_amd import synth
    X = data()
    n, m = X.shape
    wh = synth.random_state(n, m, 3, HY1, seed=2)
    eng = NumpyPhaseEngine(X, 3)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    ref = wh
    hy = {"aw": 1.4, "bw": 0.9, "ah": 0.6, "bh": 1.7}
    for _ in range(5):
        lkh, stats = eng.step(hy)
        ref = O.update_dense(X, ref, hy)
        assert abs(lkh / ref["lkh"] - 1) < 1e-11
        assert np.allclose(stats, (np.mean(np.log(ref["lw"])), np.mean(np.log(ref["lh"])), np.mean(ref["ew"]), np.mean(ref["eh"])), rtol=1e-11)
    got = eng.get_state()
    for k in ("lw", "lh", "ew", "eh", "dw", "dh"):
        assert relerr(got[k], ref[k]) < 1e-11, k


def test_hyper_update_matches_oracle_restatement():
    from ccfindr_amd import bayesian
    X = data()
    n, m = X.shape
    rng = np.random.default_rng(0)
    wh = O.update_dense(X, O.vb_init_random(n, m, 3, HY1, rng), HY1)
    for flags in ((True,) * 4, (True, False, False, True), (False, True, True, False), (False,) * 4):
        a = bayesian.hyper_update(flags, wh, HY1, Niter=100, Tol=1e-3)
        b = O.hyper_update(flags, wh, HY1, Niter=100, Tol=1e-3)
        for k in a:
            assert a[k] == pytest.approx(b[k], rel=1e-12)
        stats = (np.mean(np.log(wh["lw"])), np.mean(np.log(wh["lh"])), np.mean(wh["ew"]), np.mean(wh["eh"]))
        c = bayesian.hyper_update(flags, stats, HY1, Niter=100, Tol=1e-3)
        for k in a:
            assert c[k] == pytest.approx(a[k], rel=1e-13)
    # bh is overwritten whenever any flag is on, whatever flag 4 says (R/bayesian.R:50-51)
    assert bayesian.hyper_update((True, False, False, False), wh, HY1)["bh"] == pytest.approx(np.mean(wh["eh"]))


def test_vb_factorize_loop_matches_oracle_loop():
    """Iteration count, the lk0 lag on the convergence break, hyper trajectory, selected factors."""
    import ccfindr_amd as C
    from ccfindr_amd import bayesian
    X = data()
    n, m = X.shape
    res = C.vb_factorize(X, ranks=[2, 3], nrun=2, verbose=0, Itmax=300, Tol=1e-5, seed=5,
                         engine_factory=lambda M, r: NumpyPhaseEngine(X, r))
    assert res.ranks == [2, 3]
    for k, rank in enumerate(res.ranks):
        best = None
        for irun in (1, 2):
            rng = bayesian._bundle_rng({"seed": 5}, irun, rank)
            wh0 = bayesian.vb_init(n, m, None, rank, HY1, "random", rng=rng)
            upd = lambda wh, hy, fud: O.update_dense(X, wh, hy, fud)
            whO, hyO, lk0O, itO, trace = O.vb_iterate(upd, wh0, dict(HY1), Itmax=300, Tol=1e-5)
            if best is None or lk0O > best[0]:
                best = (lk0O, itO, hyO, whO)
        assert res.measure["lml"][k] == pytest.approx(best[0], rel=1e-9)
        assert res.nsteps[k] == best[1]
        for key in ("aw", "bw", "ah", "bh"):
            assert res.measure[key][k] == pytest.approx(best[2][key], rel=1e-8)
        assert relerr(res.basis[k], best[3]["ew"]) < 1e-7
        assert relerr(res.dcoeff[k], np.sqrt(best[3]["dh"])) < 1e-7          # sd, not variance (R/bayesian.R:382-383)


def test_guards_mirror_the_reference():
    import ccfindr_amd as C
    X = data()
    Xz = X.copy(); Xz[3, :] = 0
    with pytest.raises(ValueError, match="empty rows"):
        C.vb_factorize(Xz, ranks=2, verbose=0, engine_factory=lambda M, r: NumpyPhaseEngine(Xz, r))
    Xc = X.copy(); Xc[:, 7] = 0
    with pytest.raises(ValueError, match="empty columns"):
        C.vb_factorize(Xc, ranks=2, verbose=0, engine_factory=lambda M, r: NumpyPhaseEngine(Xc, r))
    with pytest.raises(ValueError, match="SVD initializer"):
        C.vb_factorize(X, ranks=2, nrun=2, initializer="svd2", verbose=0)
    with pytest.raises(ValueError, match="Unknown initializer"):
        C.vb_factorize(X, ranks=2, initializer="nope", verbose=0, engine_factory=lambda M, r: NumpyPhaseEngine(X, r))
    with pytest.raises(ValueError, match="Rank exceeded"):
        C.vb_factorize(X, ranks=X.shape[0] + 5, verbose=0, engine_factory=lambda M, r: NumpyPhaseEngine(X, r))   # <= ncol, > nrow
    # ranks above the number of cells are silently dropped (R/bayesian.R:249)
    Xs = np.asfortranarray(X[:12, :5] + 1.0)
    res = C.vb_factorize(Xs, ranks=[2, 9], verbose=0, Itmax=3, seed=1, engine_factory=lambda M, r: NumpyPhaseEngine(Xs, r))
    assert res.ranks == [2]


def test_nan_breaks_the_loop():
    """is.na(lkh) ends the iteration (R/bayesian.R:345); lk0 keeps the last finite value."""
    import ccfindr_amd as C

    class NaNAfter3(NumpyPhaseEngine):
        calls = 0

        def step(self, hyper, fudge=C.EPS):
            NaNAfter3.calls += 1
            lkh, st = super().step(hyper, fudge)
            return (float("nan"), st) if NaNAfter3.calls >= 3 else (lkh, st)

    X = data()
    res = C.vb_factorize(X, ranks=2, verbose=0, Itmax=50, seed=3, engine_factory=lambda M, r: NaNAfter3(X, r))
    assert res.nsteps == [3] and math.isfinite(res.measure["lml"][0])


def test_uniform_basis_column_stops_the_rank_scan():
    """A constant column of E[W] warns and, with unif.stop, ends the scan (R/bayesian.R:368-378)."""
    import ccfindr_amd as C

    class Uniform(NumpyPhaseEngine):
        def get_state(self, names=()):
            s = super().get_state(names)
            if self.rank >= 3:
                s["ew"][:, 1] = 0.25
            return s

    X = data()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = C.vb_factorize(X, ranks=[2, 3, 4], verbose=0, Itmax=5, seed=3, engine_factory=lambda M, r: Uniform(X, r))
    assert res.ranks == [2]
    msgs = [str(x.message) for x in w]
    assert any("Rank 3 row/column 2 constant." in s for s in msgs) and any("Rank scan stopped for rank >= 3" in s for s in msgs)
    with pytest.raises(RuntimeError, match="Rerun with lower ranks"):
        C.vb_factorize(X, ranks=[3, 4], verbose=0, Itmax=5, seed=3, engine_factory=lambda M, r: Uniform(X, r))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = C.vb_factorize(X, ranks=[2, 3], verbose=0, Itmax=5, seed=3, unif_stop=False, engine_factory=lambda M, r: Uniform(X, r))
    assert res.ranks == [2, 3]


def test_svd2_initialiser_shapes_and_scale():
    from ccfindr_amd import bayesian
    X = data()[:6, :40]                                  # min(nrow, ncol) / 2 <= rank: the reference's full-SVD branch
    n, m = X.shape                                       # (:151-152); the irlba branch runs on the device (GPU tests)
    wh = bayesian.vb_init(n, m, X, 3, HY1, "svd2")
    assert wh["lw"].shape == (n, 3) and wh["lh"].shape == (3, m) and (wh["lw"] >= 0).all() and (wh["lh"] >= 0).all()
    u, d, vt = np.linalg.svd(X, full_matrices=False)
    scale = HY1["bh"] / np.mean(np.abs(np.diag(d[:3]) @ vt[:3]))
    assert np.allclose(wh["lw"], np.abs(u[:, :3]) / scale) and np.allclose(wh["lh"], np.abs(np.diag(d[:3]) @ vt[:3]) * scale)
    assert np.mean(wh["lh"]) == pytest.approx(HY1["bh"])                      # R/bayesian.R:157-158
    assert not wh["dw"].any() and not wh["dh"].any()


def test_lpt_schedule_and_partition():
    from ccfindr_amd import parallel
    tasks, costs = parallel.sweep_tasks(range(2, 21), 1)
    sched = parallel.lpt_schedule(costs, 8)
    assert sorted(t for w in sched for t in w) == list(range(len(tasks)))
    loads = [sum(costs[t] for t in w) for w in sched]
    assert max(loads) <= sum(costs) / 8 + max(costs)                          # LPT bound
    assert parallel.lpt_schedule(costs, 8) == sched                           # deterministic
    parts = parallel.cell_partition(50001, 8)
    assert parts[0][0] == 0 and parts[-1][1] == 50001 and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    assert max(e - b for b, e in parts) - min(e - b for b, e in parts) <= 1


def test_vb_init_svd_matches_oracle_restatement_and_is_nonnegative():
    """initializer = 'svd' (reference R/bayesian.R:116-149), quirks kept."""
    from ccfindr_amd import bayesian
    from oracle import vbnmf_oracle as O
    rng = np.random.default_rng(12)
    X = rng.poisson(1.2, size=(40, 55)).astype(np.float64)
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    for rank in (2, 5):
        got = bayesian.vb_init(40, 55, X, rank, hy, "svd")
        want = O.vb_init_svd(X, rank)
        for k in ("w", "h", "lw", "lh", "ew", "eh", "dw", "dh"):
            assert np.allclose(got[k], want[k], rtol=1e-13, atol=0), k
        assert (got["w"] >= 0).all() and (got["h"] >= 0).all()
        # the leading pair is the Perron pair of a non-negative matrix: strictly positive, and alone it is the best rank-1 fit
        u, d, vt = np.linalg.svd(X, full_matrices=False)
        assert np.allclose(np.outer(got["w"][:, 0], got["h"][0]), d[0] * np.outer(u[:, 0], vt[0]), rtol=1e-12, atol=1e-12)
    with pytest.raises(ValueError, match="rank >= 2"):
        bayesian.vb_init(40, 55, X, 1, hy, "svd")


def test_cluster_id_is_one_based_first_maximum():
    import ccfindr_amd as C
    res = C.VBResult(ranks=[2, 3], coeff=[np.array([[1.0, 0.2], [0.5, 0.9]]), np.array([[0.1, 3.0, 2.0], [0.7, 3.0, 2.0], [0.7, 1.0, 5.0]])])
    assert C.cluster_id(res, 2).tolist() == [1, 2]
    assert C.cluster_id(res, 3).tolist() == [2, 1, 3]
    with pytest.raises(IndexError):
        C.cluster_id(res, 4)


def test_batched_restarts_driver_equals_the_run_by_run_driver(monkeypatch):
    """vb_factorize's batched form (bayesian.vb_iterate_batched: the runs of a rank stepped together, rank by rank; the device
    side is engine.run_batch / vbnmf_batch_run, tests/test_gpu_batch_run.py) against vb_iterate run by run (reference
    R/bayesian.R:260-261): same records, same best runs, and a run's scan still ends at ITS first rank with a constant column
    under unif.stop (:373-377) while the other runs go on.  The batch runner here is the host-stepped loop on the numpy engine."""
    import ccfindr_amd as C
    from ccfindr_amd import bayesian as B, engine as E

    def host_loop_batch(engines, hypers, Itmax=10000, Tol=1e-5, n0=10, dn=1, flags=(True,) * 4, fudge=C.EPS, history=False):
        outs = []
        for eng, hyper in zip(engines, hypers):
            hyper, lk0, it = dict(hyper), 0.0, 0
            for it in range(1, Itmax + 1):                                       # R/bayesian.R:337-352
                lkh, stats = eng.step(hyper, fudge)
                if it > n0 and it % dn == 0:
                    hyper = B.hyper_update(list(flags), stats, hyper, Niter=100, Tol=1e-3)
                if math.isnan(lkh):
                    break
                if it > 1 and it > n0 and lkh >= lk0 and abs(1 - lkh / lk0) < Tol:
                    break
                lk0 = lkh
            outs.append({"it": it, "lk0": lk0, "lkh": lkh, "reason": 0, "hyper": hyper, "history": None})
        return outs

    monkeypatch.setattr(E, "run_batch", host_loop_batch)
    X = data()

    class UniformInRunTwo(NumpyPhaseEngine):                                     # run 2's rank 3 comes out with a constant column
        made = 0

        def __init__(self, X, rank):
            super().__init__(X, rank)
            UniformInRunTwo.made += 1

        def set_state(self, lw, lh, eh):
            super().set_state(lw, lh, eh)
            self.mark = float(lw[0, 0])

        def get_state(self, names=()):
            s = super().get_state(names)
            if self.rank == 3 and self.mark == UniformInRunTwo.poison:
                s["ew"][:, 1] = 0.25
            return s

    ranks, nrun = [2, 3, 4], 3
    kw = dict(ranks=ranks, nrun=nrun, verbose=0, initializer="random", Itmax=12, hyper_update=(True,) * 4, gamma_a=1, gamma_b=1, Tol=1e-5,
              hyper_update_n0=3, hyper_update_dn=1, fudge=None, unif_stop=True, seed=5, device=0)
    factory = lambda M, r: UniformInRunTwo(X, r)                                 # noqa: E731
    # the first element of run 2 / rank 3's start identifies that unit whatever the order the units are run in
    rng = B._bundle_rng({"seed": 5}, 2, 3)
    UniformInRunTwo.poison = float(B.vb_init(X.shape[0], X.shape[1], X, 3, hyper={"aw": 1.0, "ah": 1.0, "bw": 1.0, "bh": 1.0},
                                             initializer="random", rng=rng)["lw"][0, 0])
    results = []
    for mode in ("run by run", "rank by rank", "across ranks", "across ranks, everything at once"):
        bundle = B.make_bundle(X, engine_factory=factory, **kw)
        bundle["device_loop"] = False
        bundle["concurrent"] = 1
        bundle["engines"] = {}
        B.plan_geometry(bundle, 1)
        if mode.startswith("across"):
            bundle["pad_rank"] = 4                                               # (what lets a batch span ranks; the numpy engine has no width)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            vb = ([B.vb_iterate(irun, bundle) for irun in range(1, nrun + 1)] if mode == "run by run" else
                  B.vb_iterate_batched(bundle, {"rank by rank": 2, "across ranks": 4}.get(mode, 64)))
        B._close_engines(bundle)
        results.append(vb)
    for other in results[1:]:
        for a, b in zip(results[0], other):
            assert a["rdat"] == b["rdat"] and a["nsteps"] == b["nsteps"] and a["hyperp"] == b["hyperp"]
            for k in a["wdat"]:
                assert np.array_equal(a["wdat"][k], b["wdat"][k]) and np.array_equal(a["hdat"][k], b["hdat"][k])
        assert other[1]["rdat"][1] == -math.inf and other[1]["rdat"][2] == -math.inf        # run 2 stopped at rank 3 (its rank 4, run
        assert other[0]["rdat"][2] > -math.inf and other[2]["rdat"][2] > -math.inf          # ahead across ranks, is dropped); 1 and 3 went on
    # what decides whether vb_factorize batches, and on which grids
    bundle = B.make_bundle(X, **dict(kw, nrun=5))
    bundle.update(device_loop=True, concurrent=1)
    assert B.batch_eligible(bundle, None, False) == 5 and B.batch_eligible(bundle, 1) == 1 and B.batch_eligible(bundle, 3, False) == 3
    assert B.batch_eligible(bundle, None) == 15 and B.batch_across_ranks(bundle, None) and not B.batch_across_ranks(dict(bundle, ranks=[3]), None)
    assert B.batch_eligible(dict(bundle, nrun=40), None) == 16 and B.batch_eligible(dict(bundle, nrun=1), None) == 3      # (nrun = 1: the ranks)
    assert B.batch_eligible(dict(bundle, nrun=1), None, False) == 1 and B.batch_eligible(dict(bundle, nrun=1, ranks=[3]), None) == 1
    assert E.auto_batch(3e5, 20) == 16 and E.auto_batch(2e6, 20) == 8 and E.auto_batch(1.5e7, 20) == 4 and E.auto_batch(5e7, 20) == 1 and E.auto_batch(3e5, 3) == 3
    assert B.batch_eligible(dict(bundle, ranks=[2, 17]), None) == 1 and B.batch_eligible(dict(bundle, concurrent=4), None) == 1
    with pytest.raises(ValueError):
        B.batch_eligible(dict(bundle, nrun=1, ranks=[3]), 4)
    assert E.batch_grid(1) == (256, 256) and E.batch_grid(8) == (32, 32) and E.batch_grid(5) == (48, 48) and E.batch_grid(64) == (8, 8)
