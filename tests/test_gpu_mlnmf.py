"""GPU parity of the maximum-likelihood NMF step: the HIP engine, through the C ABI,
against the CPU restatements of reference R/factorize.R:2-27 (nmf_updateR) and :40-49 (likelihood).

Tolerances (fp64): one step on identical inputs -- factors max relative error <= 1e-12, likelihood relative
error <= 1e-10; trajectories of tens of steps -- 1e-9.  Parity is unpinned (the reference holds no outputs for
this path and R is absent): the checker is oracle/mlnmf_oracle.py (dense literal) and mlnmf_oracle.c (stored
entries), which tests/test_oracle_mlnmf.py holds to each other.
"""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def counts(n, m, lam, seed):
    rng = np.random.default_rng(seed)
    X = rng.poisson(lam, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1      # no empty rows
    X[rng.integers(0, n, m), np.arange(m)] += 1      # no empty columns
    return np.asfortranarray(X)


def uniform_state(n, m, r, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(size=(n, r)), rng.uniform(size=(r, m))


def check(got, want_w, want_h, want_lk, tol_f=1e-12, tol_l=1e-10):
    assert relerr(got["ew"], want_w) <= tol_f, relerr(got["ew"], want_w)
    assert relerr(got["eh"], want_h) <= tol_f, relerr(got["eh"], want_h)
    assert abs(got["lk"] / want_lk - 1) <= tol_l, (got["lk"], want_lk)


@pytest.mark.parametrize("n,m,r,lam", [(64, 96, 4, 1.5), (200, 500, 3, 0.8), (300, 400, 10, 0.05), (130, 70, 1, 1.0),
                                       (97, 211, 7, 0.3), (50, 60, 20, 2.0), (40, 45, 32, 1.0), (1500, 2300, 5, 0.1)])
def test_single_step_dense_matches_literal_oracle(n, m, r, lam):
    import ccfindr_amd as C
    from oracle import mlnmf_oracle as O
    X = counts(n, m, lam, seed=n + m + r)
    w, h = uniform_state(n, m, r, seed=11)
    got = C.nmf_update(X, w, h)
    want = O.nmf_update_literal(X, w, h)
    check(got, want["ew"], want["eh"], O.likelihood_literal(X, want["ew"], want["eh"]))


def test_sparse_input_equals_dense_input_bitwise():
    import ccfindr_amd as C
    X = counts(120, 340, 0.2, seed=3)
    w, h = uniform_state(120, 340, 6, seed=4)
    a = C.nmf_update(X, w, h)
    b = C.nmf_update(sp.csc_matrix(X), w, h)
    c = C.nmf_update(sp.csr_matrix(X), w, h)
    for k in ("ew", "eh"):
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k])
    assert a["lk"] == b["lk"] == c["lk"]


def test_gamma_prior_variant():
    """prior = TRUE (R/factorize.R:10-13, :19-22)."""
    import ccfindr_amd as C
    from oracle import mlnmf_oracle as O
    X = counts(90, 140, 0.6, seed=21)
    w, h = uniform_state(90, 140, 5, seed=22)
    got = C.nmf_update(X, w, h, prior=True, gamma_a=2.5, gamma_b=0.7)
    want = O.nmf_update_literal(X, w, h, True, 2.5, 0.7)
    check(got, want["ew"], want["eh"], O.likelihood_literal(X, want["ew"], want["eh"]))


def test_non_integer_and_large_counts_use_wide_layout():
    import ccfindr_amd as C
    from oracle import mlnmf_oracle as O
    X = counts(80, 150, 0.7, seed=5)
    X = X * (np.median(X.sum(axis=0)) / X.sum(axis=0))[None, :]
    X[3, 4] = 70000.25
    w, h = uniform_state(80, 150, 5, seed=9)
    got = C.nmf_update(X, w, h)
    want = O.nmf_update_literal(X, w, h)
    check(got, want["ew"], want["eh"], O.likelihood_literal(X, want["ew"], want["eh"]))


def test_clip_at_eps():
    """Entries driven to 0 are clipped at .Machine$double.eps (R/factorize.R:15,24): a gene expressed in one cell
    only pulls its row of w towards 0 in the components that cell does not use."""
    import ccfindr_amd as C
    from oracle import mlnmf_oracle as O
    rng = np.random.default_rng(8)
    X = counts(30, 40, 0.9, seed=8)
    w = rng.uniform(size=(30, 3))
    h = rng.uniform(size=(3, 40))
    h[1, :] = 0.0                                  # a dead component: up = 0 -> clipped
    got = C.nmf_update(X, w, h)
    want = O.nmf_update_literal(X, w, h)
    assert np.any(want["eh"] == O.EPS)
    check(got, want["ew"], want["eh"], O.likelihood_literal(X, want["ew"], want["eh"]))
    assert np.array_equal(got["eh"] == O.EPS, want["eh"] == O.EPS)


def test_engine_trajectory_and_loaded_state_likelihood():
    import ccfindr_amd as C
    from oracle import mlnmf_oracle as O
    n, m, r = 180, 260, 6
    X = counts(n, m, 0.4, seed=31)
    w, h = uniform_state(n, m, r, seed=32)
    eng = C.VBEngine(C.CountMatrix(X), r)
    eng.ml_set_state(w, h)
    assert abs(eng.ml_likelihood() / O.likelihood_literal(X, w, h) - 1) <= 1e-10
    lks = []
    for _ in range(40):
        lk = eng.ml_step()
        o = O.nmf_update_literal(X, w, h)
        w, h = o["ew"], o["eh"]
        lks.append((lk, O.likelihood_literal(X, w, h)))
    st = eng.ml_get_state()
    eng.close()
    assert relerr(st["ew"], w) <= 1e-9 and relerr(st["eh"], h) <= 1e-9
    assert max(abs(a / b - 1) for a, b in lks) <= 1e-9
    assert all(b[0] >= a[0] - 1e-13 * abs(a[0]) for a, b in zip(lks, lks[1:]))     # likelihood never decreases


def test_ml_and_vb_states_exclude_each_other():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    X = counts(60, 80, 0.5, seed=41)
    eng = C.VBEngine(C.CountMatrix(X), 4)
    with pytest.raises(C.VBNMFError):
        eng.ml_step()
    w, h = uniform_state(60, 80, 4, seed=42)
    eng.ml_set_state(w, h)
    with pytest.raises(C.VBNMFError):
        eng.step(hy)                                # the VB state was never loaded
    wh = synth.random_state(60, 80, 4, hy, seed=43)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    with pytest.raises(C.VBNMFError):
        eng.ml_step()                               # loading a VB state dropped the ML one
    eng.close()


def test_medium_sparse_against_stored_entries_oracle():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import mlnmf_oracle as O
    X = synth.fill_empty(synth.simulate_data(3000, [1000] * 5, alpha0=0.1, seed=7, nfactor=1), seed=7)
    n, m = X.shape
    r = 8
    w, h = uniform_state(n, m, r, seed=51)
    eng = C.VBEngine(C.CountMatrix(X), r)
    eng.ml_set_state(w, h)
    S = X.tocsc()
    for _ in range(5):
        lk = eng.ml_step()
        o = O.update_csc(n, m, S.indptr, S.indices, S.data, w, h, nthreads=8)
        w, h = o["ew"], o["eh"]
        assert abs(lk / o["lk"] - 1) <= 1e-10
    st = eng.ml_get_state()
    eng.close()
    assert relerr(st["ew"], w) <= 1e-10 and relerr(st["eh"], h) <= 1e-10


def test_factorize_matches_oracle_loop():
    """factorize() end to end (likelihood criterion) against the oracle's loop on the same uniform draws."""
    import ccfindr_amd as C
    from oracle import mlnmf_oracle as O
    X = counts(70, 110, 0.8, seed=61)
    res = C.factorize(X, ranks=[2, 3], nrun=3, verbose=0, Tol=1e-6, Itmax=500, seed=5)
    rng = np.random.default_rng(5)
    for irank, rank in enumerate([2, 3]):
        best, conav, steps = None, 0.0, []
        for irun in range(3):
            wh = O.init(70, 110, rank, rng)
            run = O.factorize_run(lambda w, h: O.nmf_update_literal(X, w, h), X, wh, Itmax=500, Tol=1e-6)
            steps.append(run["it"])
            conav = conav + O.connectivity(run["eh"])
            if best is None or run["lk"] > best["lk"]:
                best = run
        assert res.nsteps[irank] == steps
        assert abs(res.measure["likelihood"][irank] / best["lk"] - 1) <= 1e-9
        assert relerr(res.basis[irank], best["ew"]) <= 1e-7 and relerr(res.coeff[irank], best["eh"]) <= 1e-7
        assert abs(res.measure["dispersion"][irank] - O.dispersion(conav / 3, 110)) <= 1e-12
        assert abs(res.measure["cophenetic"][irank] - O.cophenet(conav / 3, 110)) <= 1e-9


# ---- committed golden vectors (tests/golden/ml_*.npz, tests/golden/make_golden_ml.py) ----
import glob
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "ml_step_*.npz"))), ids=lambda p: os.path.basename(p)[8:-4])
def test_step_golden(path):
    import ccfindr_amd as C
    z = np.load(path)
    got = C.nmf_update(z["X"], z["w0"], z["h0"], prior=bool(z["prior"]), gamma_a=float(z["gamma"][0]), gamma_b=float(z["gamma"][1]))
    check(got, z["ew"], z["eh"], float(z["lk"]))


def test_trajectory_and_stop_golden():
    import ccfindr_amd as C
    z = np.load(os.path.join(GOLD, "ml_traj_120x200_r3.npz"))
    eng = C.VBEngine(C.CountMatrix(z["X"]), 3)
    eng.ml_set_state(z["w0"], z["h0"])
    lk = [eng.ml_step() for _ in range(60)]
    st = eng.ml_get_state()
    assert max(abs(a / b - 1) for a, b in zip(lk, z["lk"])) <= 1e-9
    assert relerr(st["ew"], z["ew60"]) <= 1e-9 and relerr(st["eh"], z["eh60"]) <= 1e-9
    # the loop's stopping rule (R/factorize.R:211): same iteration as the oracle's run
    eng.ml_set_state(z["w0"], z["h0"])
    lkold, it = -np.inf, 0
    for it in range(1, 2001):
        lk0 = eng.ml_step()
        if abs(lkold - lk0) < float(z["tol"]) * abs(lkold):
            break
        lkold = lk0
    eng.close()
    assert it == int(z["it"]) and abs(lk0 / float(z["lk_stop"]) - 1) <= 1e-9


def test_pbmc_sample_golden_sparse_input():
    """The reference's bundled PBMC sample as dgCMatrix slots, never densified."""
    import ccfindr_amd as C
    z = np.load(os.path.join(GOLD, "ml_pbmc_extdata_r5.npz"))
    d = np.load(os.path.join(GOLD, "pbmc_extdata_r5.npz"))
    n, m = int(d["n"]), int(d["m"])
    X = sp.csc_matrix((d["data"].astype(np.float64), d["indices"], d["indptr"]), shape=(n, m))
    eng = C.VBEngine(C.CountMatrix(X), 5)
    eng.ml_set_state(z["w0"], z["h0"])
    lk = [eng.ml_step() for _ in range(20)]
    st = eng.ml_get_state()
    eng.close()
    assert max(abs(a / b - 1) for a, b in zip(lk, z["lk"])) <= 1e-10
    assert relerr(st["ew"], z["ew20"]) <= 1e-10 and relerr(st["eh"], z["eh20"]) <= 1e-10


def test_c3_full_size_against_stored_entries_oracle_and_monotone():
    """BASELINE.json's C3 (20 000 x 50 000, ~5 % stored, rank 10): two steps against the OpenMP oracle, then the
    size-independent property of the multiplicative updates -- the likelihood never decreases."""
    import os as _os
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import mlnmf_oracle as O
    n, m, r, k = 20000, 50000, 10, 10
    depth = np.round(np.random.default_rng(3).lognormal(np.log(1500.0), 0.3, size=m)).astype(np.int64)
    X = synth.fill_empty(synth.simulate_data(n, [m // k] * k, alpha0=0.065, seed=3, depth=depth), seed=3)
    w, h = uniform_state(n, m, r, seed=71)
    eng = C.VBEngine(C.CountMatrix(X), r)
    eng.ml_set_state(w, h)
    S = X.tocsc()
    nt = min(16, len(_os.sched_getaffinity(0)))
    for _ in range(2):
        lk = eng.ml_step()
        o = O.update_csc(n, m, S.indptr, S.indices, S.data, w, h, nthreads=nt)
        w, h = o["ew"], o["eh"]
        assert abs(lk / o["lk"] - 1) <= 1e-10, (lk, o["lk"])
    st = eng.ml_get_state()
    assert relerr(st["ew"], w) <= 1e-10 and relerr(st["eh"], h) <= 1e-10
    prev = lk
    for _ in range(30):
        lk = eng.ml_step()
        assert lk >= prev - 1e-13 * abs(prev)
        prev = lk
    eng.close()


def test_cluster_ids_on_device_and_connectivity_criterion():
    """which.max over the resident coefficients (first maximum on ties, NaN skipped), and factorize() under
    criterion = 'connectivity' (R/factorize.R:198-208) against the oracle's loop."""
    import ccfindr_amd as C
    from oracle import mlnmf_oracle as O
    X = counts(50, 90, 0.8, seed=71)
    w, h = uniform_state(50, 90, 4, seed=72)
    h[:, 3] = [0.5, 0.9, 0.9, 0.1]                    # a tie: the first maximum wins
    h[:, 4] = [np.nan, 0.2, 0.1, 0.05]                # NaN never wins
    eng = C.VBEngine(C.CountMatrix(X), 4)
    eng.ml_set_state(w, h)
    ids = eng.cluster_ids()
    want = np.array([np.nanargmax(h[:, j]) + 1 for j in range(90)])
    assert ids.dtype == np.int32 and np.array_equal(ids, want) and ids[3] == 2 and ids[4] == 2
    eng.close()
    res = C.factorize(X, ranks=3, nrun=2, verbose=0, criterion="connectivity", ncnn_step=12, Itmax=400, seed=9)
    rng = np.random.default_rng(9)
    steps = []
    for irun in range(2):
        wh = O.init(50, 90, 3, rng)
        run = O.factorize_run(lambda a, b: O.nmf_update_literal(X, a, b), X, wh, Itmax=400, criterion="connectivity", ncnn_step=12)
        steps.append(run["it"])
    assert res.nsteps[0] == steps


def test_device_driven_ml_loop_equals_host_stepped():
    """vbnmf_engine_ml_run (factorize()'s likelihood-criterion loop on the device) against the same loop stepped from
    the host: same iteration count, same likelihood history, same factors; and the golden stop iteration."""
    import ccfindr_amd as C
    z = np.load(os.path.join(GOLD, "ml_traj_120x200_r3.npz"))
    eng = C.VBEngine(C.CountMatrix(z["X"]), 3)
    eng.ml_set_state(z["w0"], z["h0"])
    run = eng.ml_run(Itmax=2000, Tol=float(z["tol"]), history=True)
    dev = eng.ml_get_state()
    assert run["reason"] == 2 and run["it"] == int(z["it"]) and abs(run["lk"] / float(z["lk_stop"]) - 1) <= 1e-9
    assert eng.ml_likelihood() == run["lk"] and run["history"].shape == (run["it"],) and run["history"][-1] == run["lk"]
    eng.ml_set_state(z["w0"], z["h0"])
    host = [eng.ml_step() for _ in range(run["it"])]
    st = eng.ml_get_state()
    assert np.array_equal(np.array(host), run["history"])            # the same kernels in the same order: bit for bit
    assert np.array_equal(st["ew"], dev["ew"]) and np.array_equal(st["eh"], dev["eh"])
    # Itmax reached before convergence
    eng.ml_set_state(z["w0"], z["h0"])
    run = eng.ml_run(Itmax=7, Tol=0.0)
    assert run["reason"] == 4 and run["it"] == 7 and run["lk"] == host[6]
    # the step path works again after a run
    assert eng.ml_step() == host[7]
    eng.close()
    # (batch=1: one restart at a time on the default grids, as the host-stepped loop runs them; a batch sits on smaller grids)
    a = C.factorize(z["X"], ranks=3, nrun=2, verbose=0, Tol=1e-6, Itmax=600, seed=3, device_loop=True, batch=1)
    b = C.factorize(z["X"], ranks=3, nrun=2, verbose=0, Tol=1e-6, Itmax=600, seed=3, device_loop=False)
    assert a.nsteps == b.nsteps and a.measure == b.measure and np.array_equal(a.basis[0], b.basis[0])
