"""GPU: what sits either side of the update path on the device: the Gamma-random initial state
of vb_init (reference R/bayesian.R:111-115) and the connectivity change count of factorize() (R/factorize.R:198-208)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _matrix(n, m, seed):
    rng = np.random.default_rng(seed)
    X = rng.poisson(0.4, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1
    X[rng.integers(0, n, m), np.arange(m)] += 1
    return X


@pytest.mark.parametrize("aw,bw,ah,bh", [(1.0, 1.0, 1.0, 1.0), (0.3, 2.0, 5.0, 0.5), (12.0, 3.0, 0.05, 1.0)])
def test_device_gamma_initial_state_has_the_prior_distribution(aw, bw, ah, bh):
    """w ~ Gamma(shape aw, scale bw/aw), h ~ Gamma(ah, bh/ah): Kolmogorov-Smirnov against scipy's CDF, the moments,
    lw == ew, dw == 0, reproducibility by seed, and independence of the partitioning."""
    import ccfindr_amd as C
    from scipy import stats
    n, m, r = 700, 900, 6
    M = C.CountMatrix(_matrix(n, m, 1))
    hy = {"aw": aw, "bw": bw, "ah": ah, "bh": bh}
    eng = C.VBEngine(M, r)
    eng.random_state(hy, seed=1234567890123)
    st = eng.get_state()
    for name, a, b, f in (("w", aw, bw, st["lw"]), ("h", ah, bh, st["lh"])):
        x = f.ravel()
        assert np.all(x >= 0) and np.all(np.isfinite(x))
        ks = stats.kstest(x[x > 0], stats.gamma(a, scale=b / a).cdf)
        assert ks.pvalue > 1e-3 and ks.statistic < 2.2 / np.sqrt(x.size), (name, ks)
        assert abs(x.mean() / b - 1) < 5.0 / np.sqrt(a * x.size), (name, x.mean(), b)       # 5 sigma of the sample mean
    assert np.array_equal(st["lw"], st["ew"]) and np.array_equal(st["lh"], st["eh"])
    assert not st["dw"].any() and not st["dh"].any()
    lkh, _ = eng.step(hy)                                       # the state is primed: a step runs and is finite
    assert np.isfinite(lkh)
    eng2 = C.VBEngine(M, r)
    eng2.random_state(hy, seed=1234567890123)
    st2 = eng2.get_state(("lw", "lh"))
    assert np.array_equal(st2["lw"], st["lw"]) and np.array_equal(st2["lh"], st["lh"])
    eng2.random_state(hy, seed=99)
    assert not np.array_equal(eng2.get_state(("lw",))["lw"], st["lw"])
    # a partition draws its own columns of the same H
    part = C.VBEngine(M, r, cols=(300, 650), m_global=m)
    part.random_state(hy, seed=1234567890123)
    assert np.array_equal(part.get_state(("lh",))["lh"], st["lh"][:, 300:650])
    for e in (eng, eng2, part):
        e.close()


def test_vb_factorize_with_device_init_runs_and_is_reproducible():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.drop_empty(synth.simulate_data(200, (80, 120, 160), seed=5, sparse=True))
    a = C.vb_factorize(X, ranks=[2, 3], nrun=2, verbose=0, Itmax=200, seed=17, device_init=True)
    b = C.vb_factorize(X, ranks=[2, 3], nrun=2, verbose=0, Itmax=200, seed=17, device_init=True)
    assert a.measure == b.measure and all(np.array_equal(x, y) for x, y in zip(a.basis, b.basis))
    assert all(np.isfinite(v) for v in a.measure["lml"])


def test_connectivity_change_count_on_device_matches_the_pair_vectors():
    """cluster_changes against sum(cnn != cnn0) formed literally from the O(m^2) pair vectors (R/factorize.R:51-60, 201)."""
    import importlib
    import ccfindr_amd as C
    F = importlib.import_module("ccfindr_amd.factorize")       # (ccfindr_amd.factorize the attribute is the function)
    n, m, r = 150, 260, 5
    X = _matrix(n, m, 3)
    rng = np.random.default_rng(4)
    eng = C.VBEngine(C.CountMatrix(X), r)
    eng.ml_set_state(rng.uniform(size=(n, r)), rng.uniform(size=(r, m)))
    first, ids0 = eng.cluster_changes(want_ids=True)
    assert first is None
    prev = F.connectivity(eng.ml_get_state(("eh",))["eh"])
    assert np.array_equal(ids0 - 1, F.cluster_ids(eng.ml_get_state(("eh",))["eh"]))
    for _ in range(6):
        eng.ml_step()
        got, _ = eng.cluster_changes()
        cur = F.connectivity(eng.ml_get_state(("eh",))["eh"])
        assert got == int(np.sum(cur != prev))
        prev = cur
    eng.close()
    # and through factorize()'s connectivity criterion
    res = C.factorize(X, ranks=[3], nrun=2, verbose=0, seed=2, Itmax=120, criterion="connectivity", ncnn_step=10)
    assert np.isfinite(res.measure["likelihood"][0])
