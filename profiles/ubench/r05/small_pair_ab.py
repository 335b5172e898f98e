#!/usr/bin/env python3
"""profiles/ubench/r05/small_pair_ab.py -- C1 (200 x 500, rank 3), the reference's bundled PBMC sample size (1030 x 450,
rank 5), C2 (2 000 x 10 000 dense, rank 5) and a mid-size sparse matrix: microseconds per step of the device-driven loop
with both posterior updates in one launch (default) and with the two-launch form (VBNMF_NO_UPDATE_PAIR=1), same library,
same box, interleaved."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ccfindr_amd as C
from ccfindr_amd import synth
HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}


def rate(M, shape, r, steps, pair):
    os.environ["VBNMF_NO_UPDATE_PAIR"] = "0" if pair else "1"; os.environ["VBNMF_UPDATE_PAIR"] = "1" if pair else "0"
    n, m = shape
    eng = C.VBEngine(M, r)
    wh = synth.random_state(n, m, r, HY, seed=1000 + r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    eng.run(HY, Itmax=300, Tol=0.0, flags=(False,) * 4)
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        res = eng.run(HY, Itmax=steps, Tol=0.0, n0=10, dn=1, flags=(False,) * 4)
        best = max(best, res["it"] / (time.perf_counter() - t0))
    t0 = time.perf_counter()
    for _ in range(300):
        eng.step(HY)
    host = 300 / (time.perf_counter() - t0)
    eng.close()
    return best, host


cases = [("C1 200 x 500 rank 3", synth.drop_empty(synth.simulate_data(200, (100, 150, 250), seed=1, sparse=False)), 3),
         ("PBMC-sized 1030 x 450 rank 5", synth.fill_empty(synth.simulate_data(1030, [150] * 3, alpha0=0.3, seed=4, depth=np.full(450, 900))), 5),
         ("C2 2000 x 10000 dense rank 5", synth.fill_empty(synth.simulate_data(2000, [2000] * 5, alpha0=2.0, seed=2, depth=np.full(10000, 4000)), seed=2), 5),
         ("5000 x 20000 sparse rank 8", synth.fill_empty(synth.simulate_data(5000, [4000] * 5, alpha0=0.1, seed=3, depth=np.full(20000, 400))), 8)]
for name, X, r in cases:
    M = C.CountMatrix(X)
    for rep in range(2):
        for pair in (False, True):
            g, h = rate(M, X.shape, r, 3000, pair)
            print(f"{name:32s} rep {rep} {'one launch ' if pair else 'two launches'}: device-driven {g:9.1f} it/s = {1e6 / g:7.2f} us per step; host-stepped {h:9.1f} it/s = {1e6 / h:7.2f} us", flush=True)
    M.close()
