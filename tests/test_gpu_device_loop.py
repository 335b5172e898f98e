"""The device-driven loop (vbnmf_engine_run) against the host-stepped loop and the golden loop.

Same rules as reference R/bayesian.R:336-352; the only numerical difference is that the hyper-parameter
Newton step runs with the device's digamma/trigamma instead of the host's, so trajectories agree to
rounding (1e-9 after 100+ steps), with identical iteration counts."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HY1 = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def host_loop(eng, hyper, Itmax, Tol, n0, dn, flags):
    from ccfindr_amd import bayesian
    lk0, it, trace = 0.0, 0, []
    for it in range(1, Itmax + 1):
        lkh, stats = eng.step(hyper)
        if it > n0 and it % dn == 0:
            hyper = bayesian.hyper_update(flags, stats, hyper, Niter=100, Tol=1e-3)
        trace.append((lkh,) + tuple(stats) + tuple(hyper[k] for k in ("aw", "bw", "ah", "bh")))
        if np.isnan(lkh):
            break
        if it > 1 and it > n0 and lkh >= lk0 and abs(1 - lkh / lk0) < Tol:
            break
        lk0 = lkh
    return it, lk0, hyper, np.array(trace)


def test_golden_loop_iteration_count_and_lagging_lk0():
    import ccfindr_amd as C
    z = np.load(os.path.join(GOLD, "loop_c1_r3.npz"))
    eng = C.VBEngine(C.CountMatrix(z["X"]), 3)
    eng.set_state(z["lw0"], z["lh0"], z["eh0"])
    out = eng.run(HY1, Itmax=400, Tol=1e-5, history=True)
    assert out["it"] == int(z["it"]) and out["reason"] == 2
    assert abs(out["lk0"] / float(z["lk0"]) - 1) <= 1e-9 and abs(out["lkh"] / float(z["lkh_last"]) - 1) <= 1e-9
    assert out["lk0"] == out["history"][-2, 0] and out["lkh"] == out["history"][-1, 0]   # lk0 lags on the break
    for k, v in zip(("aw", "bw", "ah", "bh"), z["hyper"]):
        assert abs(out["hyper"][k] / float(v) - 1) <= 1e-8
    st = eng.get_state(("ew", "eh"))
    assert relerr(st["ew"], z["ew"]) <= 1e-7 and relerr(st["eh"], z["eh"]) <= 1e-7
    # the engine is usable afterwards: a host step continues from the frozen state
    lkh, _ = eng.step(out["hyper"])
    assert np.isfinite(lkh)
    eng.close()


@pytest.mark.parametrize("flags,n0,dn,Itmax,Tol", [((True,) * 4, 10, 1, 120, 1e-5), ((False,) * 4, 10, 1, 40, 1e-6),
                                                    ((True, False, True, False), 5, 3, 60, 1e-7), ((True,) * 4, 0, 2, 25, 0.0)])
def test_device_loop_equals_host_loop(flags, n0, dn, Itmax, Tol):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.drop_empty(synth.simulate_data(150, (80, 120, 60), seed=4, sparse=False))
    n, m = X.shape
    wh = synth.random_state(n, m, 4, HY1, seed=77)
    M = C.CountMatrix(X)
    a = C.VBEngine(M, 4); a.set_state(wh["lw"], wh["lh"], wh["eh"])
    b = C.VBEngine(M, 4); b.set_state(wh["lw"], wh["lh"], wh["eh"])
    out = a.run(HY1, Itmax=Itmax, Tol=Tol, n0=n0, dn=dn, flags=flags, history=True)
    it, lk0, hyper, trace = host_loop(b, dict(HY1), Itmax, Tol, n0, dn, flags)
    assert out["it"] == it
    assert out["reason"] in (2, 4) and (out["reason"] == 2 if it < Itmax else True)
    assert abs(out["lk0"] / lk0 - 1) <= 1e-9
    assert np.allclose(out["history"], trace, rtol=1e-8, atol=0)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert relerr(sa[k], sb[k]) <= 1e-7, k
    a.close(); b.close()


def test_vb_factorize_device_loop_and_host_loop_agree():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.drop_empty(synth.simulate_data(80, (60, 70, 50), seed=8, sparse=False))
    a = C.vb_factorize(X, ranks=[2, 3], nrun=2, verbose=0, Itmax=150, seed=21, device_loop=True)
    b = C.vb_factorize(X, ranks=[2, 3], nrun=2, verbose=0, Itmax=150, seed=21, device_loop=False)
    assert a.ranks == b.ranks and a.nsteps == b.nsteps
    assert np.allclose(a.measure["lml"], b.measure["lml"], rtol=1e-9)
    for x, y in zip(a.basis, b.basis):
        assert relerr(x, y) <= 1e-7


def test_nan_state_stops_the_device_loop_with_reason_1():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.drop_empty(synth.simulate_data(60, (40, 50), seed=2, sparse=False))
    n, m = X.shape
    wh = synth.random_state(n, m, 2, HY1, seed=5)
    wh["lw"][3, 1] = np.nan
    eng = C.VBEngine(C.CountMatrix(X), 2)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    out = eng.run(HY1, Itmax=50)
    assert out["it"] == 1 and out["reason"] == 1 and np.isnan(out["lkh"])
    eng.close()


def test_degenerate_hyper_newton_ends_with_a_reason_not_a_hang():
    """fudge = 0 with a tiny Gamma shape: exp(psi(alw)) underflows to 0 for entries without counts, the mean log
    statistic becomes -inf and the reference's step-halving loop (R/bayesian.R:28-35) would never end -- on the GPU
    that would be a wedged device.  The device loop must come back with reason 3 (hyper update failed, reported as
    the reference's error) or 1 (NaN evidence), and the host mirror must raise instead of spinning."""
    import ccfindr_amd as C
    from ccfindr_amd import bayesian, synth
    X = synth.drop_empty(synth.simulate_data(60, (40, 50), seed=3, sparse=False))
    n, m = X.shape
    hy = {"aw": 1e-4, "bw": 1.0, "ah": 1e-4, "bh": 1.0}
    wh = synth.random_state(n, m, 2, HY1, seed=2)
    eng = C.VBEngine(C.CountMatrix(X), 2)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    try:
        res = eng.run(hy, Itmax=40, Tol=1e-5, n0=2, dn=1, flags=(True,) * 4, fudge=0.0)
        assert res["reason"] in (1, 4) or res["it"] < 40, res        # ended by a rule, not by luck
    except RuntimeError as exc:
        assert "failed to converge" in str(exc)                       # reason 3, raised as the reference's stop()
    eng.close()
    with pytest.raises((RuntimeError, ValueError)):
        bayesian.hyper_update((True,) * 4, (-np.inf, -1.0, 0.5, 0.5), {"aw": 1e-4, "bw": 1.0, "ah": 1.0, "bh": 1.0})
