import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ccfindr_amd as C
from ccfindr_amd import synth
HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
for (n, cells, r) in ((200, (100, 150, 250), 3), (1000, (150, 150, 150), 5), (2000, (1000,) * 5, 10), (5000, (2000,) * 5, 10)):
    X = synth.drop_empty(synth.simulate_data(n, cells, seed=1, sparse=True))
    nn, m = X.shape
    eng = C.VBEngine(C.CountMatrix(X), r)
    wh = synth.random_state(nn, m, r, HY, seed=1)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    for _ in range(20): eng.step(HY)
    t0 = time.perf_counter(); out = eng.run(HY, Itmax=2000, Tol=0.0, flags=(False,) * 4); dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(500): eng.step(HY)
    dth = (time.perf_counter() - t0) / 500
    print(f"{nn} x {m} nnz {X.nnz} r {r}: device loop {dt / out['it'] * 1e6:.1f} us/step, host-stepped {dth * 1e6:.1f} us/step", flush=True)
    eng.close()
