"""The fp64 oracles against the step evaluated in 50-digit arithmetic (tests/util_mp_step.py, a third statement of
reference src/vbnmf_update.cpp:33-90): the reference holds no vectors ("parity unpinned"), so this is the nearest thing
to an exact answer the oracle can be held to.  Small cases; the bounds are those of well-conditioned fp64 evaluation."""
import numpy as np
import pytest

from util_mp_step import CASES, make_case, ml_step, step


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.mark.parametrize("case", CASES, ids=[f"n{c[0]}m{c[1]}r{c[2]}" for c in CASES])
@pytest.mark.parametrize("noninteger", [False, True])
def test_oracles_against_50_digit_step(case, noninteger):
    from oracle import vbnmf_oracle as O
    n, m, r, lam, hyper, fudge, seed = case
    X, wh = make_case(n, m, r, lam, hyper, fudge, seed, noninteger)
    want = step(X, wh, hyper, fudge)
    for name, got in (("dense C", O.update_dense(X, wh, hyper, fudge)), ("R twin", O.update_rtwin(X, wh, hyper, fudge))):
        for k in ("lw", "lh", "ew", "eh", "dw", "dh"):
            assert relerr(got[k], want[k]) <= 2e-14, (name, k, relerr(got[k], want[k]))
        assert abs(got["lkh"] / float(want["lkh"]) - 1) <= 1e-13, (name, got["lkh"], float(want["lkh"]))


@pytest.mark.parametrize("prior", [False, True])
@pytest.mark.parametrize("n,m,r,seed", [(7, 9, 3, 1), (12, 6, 2, 2), (5, 14, 4, 3)])
def test_ml_oracle_against_50_digit_step(n, m, r, seed, prior):
    from oracle import mlnmf_oracle as O
    rng = np.random.default_rng(seed)
    X, _ = make_case(n, m, r, 1.1, {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}, 0.0, seed, noninteger=bool(seed % 2))
    w, h = rng.uniform(0.05, 1.0, size=(n, r)), rng.uniform(0.05, 1.0, size=(r, m))
    ew, eh, lk = ml_step(X, w, h, prior, 1.7, 0.6)
    got = O.nmf_update_literal(X, w, h, prior, 1.7, 0.6)
    assert relerr(got["ew"], ew) <= 2e-14 and relerr(got["eh"], eh) <= 2e-14
    wh = ew @ eh
    scale = (np.abs(X * np.log(wh)).sum() + wh.sum()) / n / m
    assert abs(O.likelihood_literal(X, got["ew"], got["eh"]) - float(lk)) <= 1e-13 * scale
