"""The HIP engine against the step evaluated in 50-digit arithmetic (tests/util_mp_step.py), without the fp64 oracle in
between: dense stateless entry and the resident engine (whose evidence comes from the fused identity)."""
import numpy as np
import pytest

from util_mp_step import CASES, make_case, step

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
