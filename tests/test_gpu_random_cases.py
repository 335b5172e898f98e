"""GPU parity on randomly drawn shapes, ranks, densities, value kinds and hyper-parameters (seeded): VB step and
ML step through the stateless C ABI against the dense literal oracles.  Covers the corners the hand-picked cases
may miss: single rows / columns, all-zero rows and columns inside X, rank above min(n, m), dense and nearly empty
matrices, non-integer and > 16383 values (wide layout), tiny and large hyper-parameters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 7, 33, 64, 65, 130, 257, 400]))
    m = int(rng.choice([1, 3, 16, 63, 64, 129, 300, 600, 2100]))
    r = int(rng.integers(1, 33))
    lam = float(rng.choice([0.01, 0.05, 0.3, 1.0, 4.0]))
    X = rng.poisson(lam, size=(n, m)).astype(np.float64)
    kind = int(rng.integers(0, 4))
    if kind == 1:                                      # non-integer values
        X = X * rng.uniform(0.5, 1.5, size=(1, m))
    elif kind == 2 and X.size:                         # a value beyond the packed range
        X[rng.integers(0, n), rng.integers(0, m)] = 20000.0 + rng.integers(0, 100000)
    if n > 3 and rng.random() < 0.5:
        X[rng.integers(0, n), :] = 0.0                 # an all-zero gene
    if m > 3 and rng.random() < 0.5:
        X[:, rng.integers(0, m)] = 0.0                 # an all-zero cell
    if not X.any():
        X[0, 0] = 1.0
    hy = {k: float(rng.choice([0.05, 0.5, 1.0, 3.0, 40.0])) for k in ("aw", "bw", "ah", "bh")}
    return np.asfortranarray(X), r, hy, rng


@pytest.mark.parametrize("seed", range(24))
def test_vb_step_random_case(seed):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    X, r, hy, rng = draw(seed)
    n, m = X.shape
    wh = synth.random_state(n, m, r, hy, seed=seed)
    got = C.vbnmf_update(X, wh, hy, C.EPS)
    want = O.update_dense(X, wh, hy, C.EPS)
    for k in ("lw", "lh", "ew", "eh", "dw", "dh"):
        assert relerr(got[k], want[k]) <= 1e-12, (seed, k, relerr(got[k], want[k]))
    assert abs(got["lkh"] / want["lkh"] - 1) <= 1e-10, (seed, got["lkh"], want["lkh"])


@pytest.mark.parametrize("seed", range(24))
def test_ml_step_random_case(seed):
    import ccfindr_amd as C
    from oracle import mlnmf_oracle as O
    X, r, _, rng = draw(seed)
    n, m = X.shape
    w, h = rng.uniform(0.05, 1.0, size=(n, r)), rng.uniform(0.05, 1.0, size=(r, m))
    got = C.nmf_update(X, w, h)
    want = O.nmf_update_literal(X, w, h)
    assert relerr(got["ew"], want["ew"]) <= 1e-12 and relerr(got["eh"], want["eh"]) <= 1e-12, seed
    lk = O.likelihood_literal(X, want["ew"], want["eh"])
    # the likelihood is a difference of large sums (it is exactly 0 when w h reproduces x): hold the error to the
    # size of those sums, not to the size of the result
    wh = want["ew"] @ want["eh"]
    scale = (np.abs(X * np.log(wh)).sum() + wh.sum()) / n / m
    assert abs(got["lk"] - lk) <= 1e-11 * scale, (seed, got["lk"], lk, scale)


def _counts(kind, n, m, seed):
    rng = np.random.default_rng(seed)
    if kind == "all_ones":                  # binary matrix: every stored value is 1 -> the whole slice is the leading stretch
        X = (rng.random((n, m)) < 0.25).astype(np.float64)
    elif kind == "all_twos":                # every stored value is 2: no leading ones, the ones-or-twos stretch is everything
        X = 2.0 * (rng.random((n, m)) < 0.25).astype(np.float64)
    elif kind == "ones_and_twos":           # only ones and twos, in varying proportion per gene
        p2 = rng.uniform(0.05, 0.9, size=(n, 1))
        X = (rng.random((n, m)) < 0.3).astype(np.float64)
        X = X * (1.0 + (rng.random((n, m)) < p2))
    elif kind == "no_ones":                 # every stored value >= 2 -> the stretch is empty everywhere
        X = rng.poisson(0.4, size=(n, m)).astype(np.float64)
        X[X > 0] += 1.0
    elif kind == "ones_then_big":           # mostly ones with a heavy tail: stretches of very different length per task
        X = (rng.random((n, m)) < 0.3).astype(np.float64)
        big = rng.random((n, m)) < 0.03
        X[big] = rng.integers(2, 60, size=int(big.sum())).astype(np.float64)
    else:                                   # "mixed": counts 0..3, about half of the stored ones are ones
        X = rng.poisson(0.8, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1.0
    X[rng.integers(0, n, m), np.arange(m)] += 1.0
    return np.asfortranarray(X)


@pytest.mark.parametrize("kind", ["all_ones", "all_twos", "ones_and_twos", "no_ones", "ones_then_big", "mixed"])
@pytest.mark.parametrize("r", [3, 10, 16, 28, 30])
def test_leading_ones_stretch_against_the_oracle(kind, r):
    """The sweep's shorter loop over a slice's leading stored ones (layout slice_fast; deferred logarithm through a
    running product on the gene side) against the dense literal oracle, for matrices where that stretch is everything,
    nothing, or ragged; the ranks cover the pipelined two-buffer loop at 1024 / 768 / 512 threads and the one-buffer
    loop (padded rank >= 28).  Three resident steps, so the evidence of the running-product form enters the test."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    n, m = 700, 1100
    X = _counts(kind, n, m, seed=len(kind) * 100 + r)
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    wh = synth.random_state(n, m, r, hy, seed=r)
    eng = C.VBEngine(C.CountMatrix(X), r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    ref = wh
    for _ in range(3):
        lkh, _ = eng.step(hy)
        ref = O.update_dense(X, ref, hy, C.EPS)
        assert abs(lkh / ref["lkh"] - 1) <= 1e-10, (kind, r, lkh, ref["lkh"])
    got = eng.get_state()
    eng.close()
    for k in ("lw", "lh", "ew", "eh", "dw", "dh"):
        assert relerr(got[k], ref[k]) <= 1e-11, (kind, r, k, relerr(got[k], ref[k]))


def test_leading_ones_stretch_is_recorded_by_the_layout():
    """slice_fast of a binary matrix covers every full 8-entry trip of every full slice (CPU-side view of the layout)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import ccfindr_amd as C
    from util_layout import build_layout
    X = _counts("all_ones", 300, 500, seed=1)
    X[X > 1] = 1.0
    v = build_layout(C.CountMatrix(X), 0, 10)
    assert v["slice_fast"].sum() > 0.5 * v["slice_width"].sum()
    v = build_layout(C.CountMatrix(_counts("no_ones", 300, 500, seed=2) + 0.0), 0, 10)
    assert v["slice_fast"].sum() <= 0.05 * v["slice_width"].sum()
    X2 = 2.0 * (np.random.default_rng(3).random((300, 500)) < 0.3)
    X2[X2.sum(axis=1) == 0, 0] = 2.0; X2[0, X2.sum(axis=0) == 0] = 2.0
    v = build_layout(C.CountMatrix(np.asfortranarray(X2)), 0, 10)          # twos only: the second stretch, not the first
    assert v["slice_fast"].sum() == 0 and v["slice_fast2"].sum() > 0.4 * v["slice_width"].sum()
