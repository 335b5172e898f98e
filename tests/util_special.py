"""Accuracy check of the library's fp64 ln / digamma / lnGamma against mpmath (50 digits)."""
import ctypes

import mpmath
import numpy as np

from ccfindr_amd import _native as N

mpmath.mp.dps = 50


def grid():
    rng = np.random.default_rng(7)
    xs = np.concatenate([
        np.logspace(-8, 8, 641),                      # log-spaced sweep
        np.linspace(1.40, 1.52, 121),                 # around the root of psi (1.4616321...)
        [1.4616321449683623, 1.0, 2.0, 0.5, 3.0, 9.999999, 10.0, 10.000001],
        1.0 + rng.random(200) * 1e-3, 2.0 + (rng.random(200) - 0.5) * 1e-3,   # zeros of lnGamma
        rng.random(500) * 30.0, 10.0 ** rng.uniform(-12, 12, 500),
    ])
    return np.ascontiguousarray(xs, dtype=np.float64)


def evaluate(kind, xs, device):
    L = N.load()
    ys = np.empty_like(xs)
    fn = L.vbnmf_test_special_device if device else L.vbnmf_test_special_host
    N.check(fn(kind, xs.size, N.dptr(xs), N.dptr(ys)))
    return ys


def check_all(device):
    xs = grid()
    # ln: relative error (and absolute near 1)
    got = evaluate(0, xs, device)
    ref = np.array([float(mpmath.log(mpmath.mpf(float(x)))) for x in xs])
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
    near1 = np.abs(xs - 1) < 0.3
    assert np.max(err[~near1]) < 4e-16, np.max(err[~near1])
    assert np.max(np.abs(got - ref)[near1]) < 1.2e-16, np.max(np.abs(got - ref)[near1])
    # the sweep's table-driven ln: same bounds
    got = evaluate(6, xs, device)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
    assert np.max(err[~near1]) < 6e-16, np.max(err[~near1])
    assert np.max(np.abs(got - ref)[near1]) < 1.5e-16, np.max(np.abs(got - ref)[near1])
    # digamma: absolute error scaled by max(1, |psi|)  (only exp(psi) is consumed, src/vbnmf_update.cpp:59,63)
    got = evaluate(1, xs, device)
    ref = np.array([float(mpmath.digamma(mpmath.mpf(float(x)))) for x in xs])
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert np.max(err) < 3e-15, (np.max(err), xs[np.argmax(err)])
    # lnGamma: absolute error scaled by max(1, |lnGamma|).  It only enters the evidence as a sum of
    # (n+m)*r terms of size O(1..1e5) each, so ~1e-14 absolute per term is far inside the 1e-10
    # relative tolerance on lkh (difference of two ~17.5-sized quantities near the zeros at 1, 2).
    got = evaluate(2, xs, device)
    ref = np.array([float(mpmath.loggamma(mpmath.mpf(float(x)))) for x in xs])
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert np.max(err) < 2e-14, (np.max(err), xs[np.argmax(err)])
    # reciprocal
    got = evaluate(3, xs, device)
    assert np.max(np.abs(got * xs - 1)) < 4e-16
