"""The restarts of a rank stepped by one launch, under rocprofv3 --kernel-trace --stats: sixteen engines of rank 5 on the 1030 x 450
sample, 400 iterations, first one engine at a time (k_update2, k_sweep), then the batch (k_update2_batch, k_sweep_batch)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import ccfindr_amd as C
from ccfindr_amd import synth
X = synth.drop_empty(synth.simulate_data(1030, (150, 150, 150), seed=3, sparse=True))
M = C.CountMatrix(X)
n, m = X.shape
r, B, iters = 5, 16, 400
hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
whs = [synth.random_state(n, m, r, hy, seed=b) for b in range(B)]
for grid in (None, C.batch_grid(B)):
    engs = [C.VBEngine(M, r, grid=grid) for _ in range(B)]
    for eng, wh in zip(engs, whs):
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    t0 = time.perf_counter()
    res = [e.run(hy, Itmax=iters, Tol=0.0) for e in engs] if grid is None else C.run_batch(engs, [hy] * B, Itmax=iters, Tol=0.0)
    print("grid", grid, "seconds", time.perf_counter() - t0, flush=True)
    for e in engs:
        e.close()
