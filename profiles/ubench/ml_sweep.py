import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ccfindr_amd as C
from ccfindr_amd import synth
HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
cases = ((1000, (150, 150, 150), 5), (2000, (1000,) * 5, 10), (5000, (2000,) * 5, 10))
mats = [synth.drop_empty(synth.simulate_data(n, cells, seed=1, sparse=True)) for n, cells, r in cases]
for ml in (0, 8, 16, 32, 64, 128, 256):
    if ml: os.environ["VBNMF_MAX_LEN"] = str(ml)
    out = []
    for X, (n, cells, r) in zip(mats, cases):
        nn, m = X.shape
        eng = C.VBEngine(C.CountMatrix(X), r)
        wh = synth.random_state(nn, m, r, HY, seed=1)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        for _ in range(20): eng.step(HY)
        t0 = time.perf_counter(); o = eng.run(HY, Itmax=1500, Tol=0.0, flags=(False,) * 4); dt = time.perf_counter() - t0
        out.append(round(dt / o["it"] * 1e6, 1)); eng.close()
    print("max_len", ml or "auto", out, flush=True)
