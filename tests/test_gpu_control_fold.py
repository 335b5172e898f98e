"""The control step folded into the next step's gene-side update (csrc/kernels.h: ControlFold; the default of the
unpartitioned device-driven loop) against the separate control kernel of rounds 1-2 (VBNMF_NO_CONTROL_FOLD=1, read when
an engine is created): the arithmetic is the same statement by statement and every block of the update forms it from
the same inputs, so the two loops must agree BIT FOR BIT -- iteration count, stop reason, every history row, the
hyper-parameters, the lagging lk0, the state left behind -- for every place a run can end: after one step, inside a
queued batch, on a batch boundary, on Itmax, on convergence.  Reference loop: R/bayesian.R:336-352."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HY = {"aw": 1.1, "bw": 0.9, "ah": 0.8, "bh": 1.3}


def _engine(M, r, wh, fold):
    import ccfindr_amd as C
    old = os.environ.get("VBNMF_NO_CONTROL_FOLD")
    os.environ["VBNMF_NO_CONTROL_FOLD"] = "0" if fold else "1"
    try:
        eng = C.VBEngine(M, r)
    finally:
        if old is None:
            del os.environ["VBNMF_NO_CONTROL_FOLD"]
        else:
            os.environ["VBNMF_NO_CONTROL_FOLD"] = old
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    return eng


def _same(a, b):
    assert a["it"] == b["it"] and a["reason"] == b["reason"], (a["it"], b["it"], a["reason"], b["reason"])
    assert a["lk0"] == b["lk0"] and a["lkh"] == b["lkh"]
    assert a["hyper"] == b["hyper"]
    if a["history"] is not None:
        assert np.array_equal(a["history"], b["history"])


@pytest.fixture(scope="module")
def problem():
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.fill_empty(synth.simulate_data(400, [300, 350], alpha0=0.2, seed=8, depth=np.full(650, 90)))
    n, m = X.shape
    M = C.CountMatrix(X)
    yield M, n, m
    M.close()


@pytest.mark.parametrize("Itmax", [1, 2, 7, 8, 9, 16, 17, 41])
@pytest.mark.parametrize("flags", [(True,) * 4, (False,) * 4])
def test_fold_and_separate_control_agree_bit_for_bit_on_itmax(problem, Itmax, flags):
    from ccfindr_amd import synth
    M, n, m = problem
    r = 5
    wh = synth.random_state(n, m, r, HY, seed=2)
    runs = []
    for fold in (True, False):
        eng = _engine(M, r, wh, fold)
        out = eng.run(HY, Itmax=Itmax, Tol=0.0, n0=3, dn=2, flags=flags, history=True)
        st = eng.get_state()
        eng.close()
        runs.append((out, st))
    _same(runs[0][0], runs[1][0])
    assert runs[0][0]["it"] == Itmax and runs[0][0]["reason"] == 4
    for k in runs[0][1]:
        assert np.array_equal(runs[0][1][k], runs[1][1][k]), k


def test_fold_stops_on_convergence_where_the_separate_control_does_and_the_engine_goes_on(problem):
    """A loose tolerance ends the run inside a queued batch; afterwards: a host-stepped step, a second run, a third."""
    from ccfindr_amd import synth
    M, n, m = problem
    r = 4
    wh = synth.random_state(n, m, r, HY, seed=5)
    got = []
    for fold in (True, False):
        eng = _engine(M, r, wh, fold)
        a = eng.run(HY, Itmax=300, Tol=3e-4, n0=4, dn=1, history=True)
        assert a["reason"] == 2 and 5 < a["it"] < 300
        s1 = eng.step(a["hyper"])                                  # the state is exactly as the breaking step left it
        b = eng.run(a["hyper"], Itmax=13, Tol=0.0, n0=2, dn=1, history=True)
        c = eng.run(b["hyper"], Itmax=1, Tol=0.0, history=True)
        st = eng.get_state()
        eng.close()
        got.append((a, s1, b, c, st))
    for x, y in zip(got[0][:1] + got[0][2:4], got[1][:1] + got[1][2:4]):
        _same(x, y)
    assert got[0][1] == got[1][1]
    for k in got[0][4]:
        assert np.array_equal(got[0][4][k], got[1][4][k]), k


def test_fold_nan_evidence_breaks_with_reason_1(problem):
    """A non-finite state makes the evidence NaN: the loop ends after that step with reason 1 (R/bayesian.R:345), fold or not."""
    from ccfindr_amd import synth
    M, n, m = problem
    r = 3
    wh = synth.random_state(n, m, r, HY, seed=6)
    wh["lw"][0, :] = 0.0                                           # a whole factor row 0 with fudge = 0: X / wth is NaN there
    outs = []
    for fold in (True, False):
        eng = _engine(M, r, wh, fold)
        outs.append(eng.run(HY, Itmax=20, Tol=1e-5, fudge=0.0, flags=(False,) * 4, history=True))
        eng.close()
    assert outs[0]["reason"] == outs[1]["reason"] and outs[0]["it"] == outs[1]["it"]
    assert outs[0]["reason"] in (1, 4)
    assert np.array_equal(np.isnan(outs[0]["history"][:, 0]), np.isnan(outs[1]["history"][:, 0]))


@pytest.mark.parametrize("Itmax,Tol,prior", [(1, 0.0, False), (2, 0.0, False), (9, 0.0, True), (16, 0.0, False), (17, 0.0, False),
                                             (400, 1e-4, False), (400, 3e-4, True)])
def test_ml_loop_fold_and_separate_control_agree_bit_for_bit(problem, Itmax, Tol, prior):
    """The same for the maximum-likelihood loop of factorize() (mlnmf.h: MlFold into the H update; reference
    R/factorize.R:194-213): iteration count, stop reason, likelihood history and the factors, then a second run and a
    host-stepped step from the state the first one left."""
    M, n, m = problem
    r = 4
    rng = np.random.default_rng(12)
    w0, h0 = rng.uniform(0.1, 1.0, size=(n, r)), rng.uniform(0.1, 1.0, size=(r, m))
    outs = []
    for fold in (True, False):
        eng = _engine(M, r, {"lw": w0, "lh": h0, "eh": h0}, fold)
        eng.ml_set_state(w0, h0)
        a = eng.ml_run(Itmax=Itmax, Tol=Tol, prior=prior, gamma_a=1.3, gamma_b=0.8, history=True)
        b = eng.ml_run(Itmax=5, Tol=0.0, prior=prior, gamma_a=1.3, gamma_b=0.8, history=True)
        lk = eng.ml_step(prior=prior, gamma_a=1.3, gamma_b=0.8)
        st = eng.ml_get_state()
        eng.close()
        outs.append((a, b, lk, st["ew"], st["eh"]))
    for x, y in ((outs[0][0], outs[1][0]), (outs[0][1], outs[1][1])):
        assert x["it"] == y["it"] and x["reason"] == y["reason"] and x["lk"] == y["lk"]
        assert np.array_equal(x["history"], y["history"])
    if Tol > 0:
        assert outs[0][0]["reason"] == 2 and outs[0][0]["it"] < 400
    else:
        assert outs[0][0]["it"] == Itmax and outs[0][0]["reason"] == 4
    assert outs[0][2] == outs[1][2]
    assert np.array_equal(outs[0][3], outs[1][3]) and np.array_equal(outs[0][4], outs[1][4])


@pytest.mark.parametrize("Itmax", [5, 16, 23, 40])
def test_a_drained_stream_is_not_mistaken_for_a_lost_step(problem, Itmax):
    """with the fold, step t is reported by step t + 1's launch, so a stream that has DRAINED before the host
    looks (a stalled host thread: VBNMF_TEST_DRAIN_BEFORE_POLL synchronises the stream in front of every poll) shows
    queued - 1 completed steps while more are still to be queued.  The idle check used to call that "the device went idle
    before the queued steps finished"; the run must complete and equal the undisturbed one, for the VB and the ML loop."""
    from ccfindr_amd import synth
    M, n, m = problem
    r = 4
    wh = synth.random_state(n, m, r, HY, seed=6)
    rng = np.random.default_rng(3)
    w0, h0 = rng.uniform(size=(n, r)), rng.uniform(size=(r, m))
    outs = []
    for drain in ("0", "1"):
        os.environ["VBNMF_TEST_DRAIN_BEFORE_POLL"] = drain
        try:
            eng = _engine(M, r, wh, True)
            vb = eng.run(HY, Itmax=Itmax, Tol=0.0, n0=3, dn=1, flags=(True,) * 4, history=True)
            eng.ml_set_state(w0, h0)
            ml = eng.ml_run(Itmax=Itmax, Tol=0.0, history=True)
            eng.close()
        finally:
            del os.environ["VBNMF_TEST_DRAIN_BEFORE_POLL"]
        outs.append((vb, ml))
    _same(outs[0][0], outs[1][0])
    assert outs[0][0]["it"] == Itmax
    assert outs[0][1]["it"] == outs[1][1]["it"] == Itmax and np.array_equal(outs[0][1]["history"], outs[1][1]["history"])
