"""CPU: the product's factorize() driver (ccfindr_amd/factorize.py, mirror of reference R/factorize.R:140-320)
run on a stand-in engine, against the oracle's restatement of the same loop and measures."""
import numpy as np
import pytest

from oracle import mlnmf_oracle as O
from tests.fake_ml_engine import OracleMLEngine


def counts(n, m, lam, seed):
    rng = np.random.default_rng(seed)
    X = rng.poisson(lam, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1
    X[rng.integers(0, n, m), np.arange(m)] += 1
    return np.asfortranarray(X)


def fmod():
    import importlib
    return importlib.import_module("ccfindr_amd.factorize")


def factory(X):
    return lambda M, rank: OracleMLEngine(M.host, rank)


def oracle_factorize(X, ranks, nrun, seed, **kw):
    rng = np.random.default_rng(seed)
    n, m = X.shape
    out = []
    for rank in ranks:
        best, conav, steps = None, 0.0, []
        for irun in range(nrun):
            wh = O.init(n, m, rank, rng)
            run = O.factorize_run(lambda w, h: O.nmf_update_literal(X, w, h), X, wh, **kw)
            steps.append(run["it"])
            conav = conav + O.connectivity(run["eh"])
            if (irun == 0 or run["lk"] > best["lk"]) and not np.isnan(run["lk"]):
                best = run
        out.append((best, steps, O.dispersion(conav / nrun, m), O.cophenet(conav / nrun, m)))
    return out


@pytest.mark.parametrize("criterion", ["likelihood", "connectivity"])
def test_factorize_loop_matches_oracle(criterion):
    F = fmod()
    X = counts(40, 60, 0.9, seed=1)
    kw = dict(Itmax=300, Tol=1e-5, criterion=criterion, ncnn_step=15)
    res = F.factorize(X, ranks=[2, 4], nrun=3, verbose=0, seed=9, engine_factory=factory(X),
                      Itmax=300, Tol=1e-5, criterion=criterion, ncnn_step=15)
    want = oracle_factorize(X, [2, 4], 3, 9, **kw)
    assert res.ranks == [2, 4] and res.measure["rank"] == [2, 4]
    for i, (best, steps, disp, coph) in enumerate(want):
        assert res.nsteps[i] == steps
        assert res.measure["likelihood"][i] == best["lk"]
        assert np.array_equal(res.basis[i], best["ew"]) and np.array_equal(res.coeff[i], best["eh"])
        assert abs(res.measure["dispersion"][i] - disp) < 1e-14
        assert abs(res.measure["cophenetic"][i] - coph) < 1e-12
    assert set(res.measure) == {"rank", "likelihood", "dispersion", "cophenetic"}


def test_connectivity_changes_equals_pair_count():
    F = fmod()
    rng = np.random.default_rng(3)
    for _ in range(20):
        r = int(rng.integers(2, 6))
        a, b = rng.integers(0, r, 50), rng.integers(0, r, 50)
        iu = np.triu_indices(50, 1)
        brute = int(np.sum((a[iu[0]] == a[iu[1]]) != (b[iu[0]] == b[iu[1]])))
        assert F.connectivity_changes(a, b, r) == brute
    h = rng.uniform(size=(4, 30))
    assert np.array_equal(F.connectivity(h), O.connectivity(h))


def test_guards_and_randomize_and_store_connectivity():
    F = fmod()
    X = counts(30, 40, 0.8, seed=2)
    Z = X.copy(); Z[5, :] = 0
    with pytest.raises(ValueError, match="empty rows"):
        F.factorize(Z, engine_factory=factory(Z), verbose=0)
    Z = X.copy(); Z[:, 7] = 0
    with pytest.raises(ValueError, match="empty columns"):
        F.factorize(Z, engine_factory=factory(Z), verbose=0)
    with pytest.raises(ValueError, match="Unknown stopping criterion"):
        F.factorize(X, engine_factory=factory(X), verbose=0, criterion="other")
    res = F.factorize(X, ranks=2, nrun=2, nsmpl=3, randomize=True, verbose=0, seed=4, Itmax=50, store_connectivity=True,
                      engine_factory=factory(X))
    assert set(res.measure) == {"rank", "likelihood", "r_se", "dispersion", "d_se", "cophenetic", "c_se"}
    assert np.isfinite(res.measure["r_se"][0]) and res.metadata["nrun"] == 2
    assert res.metadata["connectivity"].shape == (40 * 39 // 2,)
    assert res.basis[0].shape == (30, 2) and res.coeff[0].shape == (2, 40)

