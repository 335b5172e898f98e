"""Host-side cost of ONE (run, rank) unit of a small matrix's sweep beside its loop: engine creation on cached layouts, the start drawn by
numpy (vb_init 'random'), set_state, get_state, the record -- what is left of a batched sweep once the loops share their launches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import ccfindr_amd as C
from ccfindr_amd import synth, bayesian as B
X = synth.drop_empty(synth.simulate_data(1030, (150, 150, 150), seed=3, sparse=True))
M = C.CountMatrix(X)
n, m = X.shape
hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
grid, pad = C.batch_grid(16), C.engine.padded_rank(9)
C.VBEngine(M, 5, grid=grid, pad_rank=pad).close()
def med(f, reps=20):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts)), r
t_create, eng = med(lambda: C.VBEngine(M, 5, grid=grid, pad_rank=pad))
rng = np.random.default_rng(1)
t_init, wh = med(lambda: B.vb_init(n, m, M, 5, hyper=hy, initializer="random", rng=rng))
t_set, _ = med(lambda: eng.set_state(wh["lw"], wh["lh"], wh["eh"]))
t_get, st = med(lambda: eng.get_state(("ew", "eh", "dw", "dh")))
t_rec, _ = med(lambda: B._unit_record(5, {"Tol": 1e-5}, dict(st), False, 0.0, hy, 1))
t_close, _ = med(lambda: C.VBEngine(M, 5, grid=grid, pad_rank=pad).close())
print(f"per unit (ms): engine creation {t_create:.2f}, creation + close {t_close:.2f}, vb_init {t_init:.2f}, set_state {t_set:.2f}, get_state {t_get:.2f}, record {t_rec:.2f}")
