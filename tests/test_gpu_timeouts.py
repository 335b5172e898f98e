"""The host's waits on the device are bounded (VBNMF_WAIT_TIMEOUT_S): a step or a device-driven loop that does not
come back in time returns VBNMF_ERR_HIP with a message naming the last completed step -- instead of spinning for ever on
hipErrorNotReady, as a partitioned run with a dead peer would.  The GPU is never hung here: a host function that sleeps
is put on the engine's stream (vbnmf_test_stream_sleep), so the queued steps are merely late."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}


def _engine(seed):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.fill_empty(synth.simulate_data(300, [200, 200], alpha0=0.2, seed=seed, depth=np.full(400, 120)))
    n, m = X.shape
    M = C.CountMatrix(X)
    eng = C.VBEngine(M, 4)
    wh = synth.random_state(n, m, 4, HY, seed=seed)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    return M, eng


@pytest.fixture
def short_limit():
    old = os.environ.get("VBNMF_WAIT_TIMEOUT_S")
    os.environ["VBNMF_WAIT_TIMEOUT_S"] = "1"
    yield
    if old is None:
        del os.environ["VBNMF_WAIT_TIMEOUT_S"]
    else:
        os.environ["VBNMF_WAIT_TIMEOUT_S"] = old


def test_host_stepped_wait_times_out_with_the_last_completed_step(short_limit):
    from ccfindr_amd import _native as N
    M, eng = _engine(3)
    lk0, _ = eng.step(HY)                                   # one step comes back well inside the limit
    assert np.isfinite(lk0)
    N.check(N.load().vbnmf_test_stream_sleep(eng._h, 3.0))
    t0 = time.perf_counter()
    with pytest.raises(N.VBNMFError) as ei:
        eng.step(HY)
    waited = time.perf_counter() - t0
    assert ei.value.code == N.ERR_HIP
    assert "timed out" in str(ei.value) and "last completed step" in str(ei.value)
    assert 0.9 < waited < 2.5, waited
    # the engine is unusable from here on, and every entry point says so AT ONCE instead of blocking on a
    # stream that may never drain (get_state's memcpy, set_state's synchronise, another step)
    t0 = time.perf_counter()
    for call in (lambda: eng.get_state(), lambda: eng.step(HY), lambda: eng.run(HY, Itmax=3),
                 lambda: eng.set_state(np.ones((eng.n, 4)), np.ones((4, eng.m)), np.ones((4, eng.m)))):
        with pytest.raises(N.VBNMFError) as e2:
            call()
        assert e2.value.code == N.ERR_STATE and "timed out earlier" in str(e2.value)
    assert time.perf_counter() - t0 < 0.5
    time.sleep(3.0)                                          # the sleeper ends, the queued step drains
    eng.close()
    M.close()


def test_device_driven_loop_times_out_instead_of_spinning(short_limit):
    from ccfindr_amd import _native as N
    M, eng = _engine(4)
    N.check(N.load().vbnmf_test_stream_sleep(eng._h, 3.0))
    t0 = time.perf_counter()
    with pytest.raises(N.VBNMFError) as ei:
        eng.run(HY, Itmax=40, Tol=0.0, flags=(False,) * 4)
    waited = time.perf_counter() - t0
    assert ei.value.code == N.ERR_HIP
    assert "timed out" in str(ei.value) and "device-driven loop" in str(ei.value)
    assert 0.9 < waited < 2.5, waited
    time.sleep(3.0)
    eng.close()
    M.close()
