import os, sys, time, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ccfindr_amd as C
from ccfindr_amd import synth
HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
def rate(X, r, steps):
    n, m = X.shape
    M = C.CountMatrix(X)
    eng = C.VBEngine(M, r)
    wh = synth.random_state(n, m, r, HY, seed=1000 + r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    eng.run(HY, Itmax=300, Tol=0.0, n0=10, dn=1, flags=(False,) * 4)
    best = 0
    for _ in range(3):
        t0 = time.perf_counter()
        res = eng.run(HY, Itmax=steps, Tol=0.0, n0=10, dn=1, flags=(False,) * 4)
        best = max(best, res["it"] / (time.perf_counter() - t0))
    eng.close(); M.close()
    return best
which = sys.argv[1]
if which == "c1":
    X = synth.drop_empty(synth.simulate_data(200, (100, 150, 250), seed=1, sparse=False)); r = 3
elif which == "c2":
    X = synth.fill_empty(synth.simulate_data(2000, [2000] * 5, alpha0=2.0, seed=2, depth=np.full(10000, 4000)), seed=2); r = 5
else:
    import scipy.io
    X = scipy.io.mmread(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/pbmc.mtx")) if False else None
v = rate(X, r, 3000)
print(which, "MAX_LEN", os.environ.get("VBNMF_MAX_LEN", "default"), "FOLD_OFF", os.environ.get("VBNMF_NO_CONTROL_FOLD", "0"), "it/s %.0f  us/step %.2f" % (v, 1e6 / v), flush=True)
