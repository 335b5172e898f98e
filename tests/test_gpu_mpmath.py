"""The HIP engine against the step evaluated in 50-digit arithmetic (tests/util_mp_step.py), without the fp64 oracle in
between: dense stateless entry and the resident engine (whose evidence comes from the fused identity)."""
import numpy as np
import pytest

from util_mp_step import CASES, make_case, ml_step, step

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.mark.parametrize("case", CASES, ids=[f"n{c[0]}m{c[1]}r{c[2]}" for c in CASES])
@pytest.mark.parametrize("noninteger", [False, True])
def test_engine_against_50_digit_step(case, noninteger):
    import ccfindr_amd as C
    n, m, r, lam, hyper, fudge, seed = case
    X, wh = make_case(n, m, r, lam, hyper, fudge, seed, noninteger)
    want = step(X, wh, hyper, fudge)
    got = C.vbnmf_update(X, wh, hyper, fudge)
    for k in ("lw", "lh", "ew", "eh", "dw", "dh"):
        assert relerr(got[k], want[k]) <= 1e-13, (k, relerr(got[k], want[k]))
    assert abs(got["lkh"] / float(want["lkh"]) - 1) <= 1e-12, (got["lkh"], float(want["lkh"]))
    M = C.CountMatrix(X)
    eng = C.VBEngine(M, r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    lkh, _ = eng.step(hyper, fudge)
    assert abs(lkh / float(want["lkh"]) - 1) <= 1e-12
    eng.close()


@pytest.mark.parametrize("prior", [False, True])
@pytest.mark.parametrize("n,m,r,seed", [(7, 9, 3, 1), (12, 6, 2, 2), (5, 14, 4, 3)])
def test_ml_engine_against_50_digit_step(n, m, r, seed, prior):
    import ccfindr_amd as C
    rng = np.random.default_rng(seed)
    X, _ = make_case(n, m, r, 1.1, {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}, 0.0, seed, noninteger=bool(seed % 2))
    w, h = rng.uniform(0.05, 1.0, size=(n, r)), rng.uniform(0.05, 1.0, size=(r, m))
    ew, eh, lk = ml_step(X, w, h, prior, 1.7, 0.6)
    got = C.nmf_update(X, w, h, prior=prior, gamma_a=1.7, gamma_b=0.6)
    assert relerr(got["ew"], ew) <= 1e-13 and relerr(got["eh"], eh) <= 1e-13
    wh = ew @ eh
    scale = (np.abs(X * np.log(wh)).sum() + wh.sum()) / n / m
    assert abs(got["lk"] - float(lk)) <= 1e-12 * scale
