"""Up to which matrix size does stepping the restarts of a rank together pay?  Per matrix: four and eight restarts one loop at a time
(default grids) against one batch (256 / B workgroups per engine), aggregate iterations per second."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import ccfindr_amd as C
from ccfindr_amd import synth

def sample(n, m, density, seed):
    rng = np.random.default_rng(seed)
    import scipy.sparse as sp
    X = sp.random(n, m, density=density, format="csc", random_state=rng, data_rvs=lambda k: rng.integers(1, 6, k).astype(np.float64))
    X = X + sp.csc_matrix((np.ones(n), (np.arange(n), rng.integers(0, m, n))), shape=(n, m)) + sp.csc_matrix((np.ones(m), (rng.integers(0, n, m), np.arange(m))), shape=(n, m))
    return sp.csc_matrix(X)

hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
for name, n, m, dens, r in (("1 000 x 450, 78 %", 1000, 450, 0.78, 5), ("2 000 x 5 000, 20 %", 2000, 5000, 0.2, 8), ("5 000 x 20 000, 5 %", 5000, 20000, 0.05, 8),
                            ("2 000 x 10 000, 75 % (C2)", 2000, 10000, 0.75, 5), ("10 000 x 30 000, 5 %", 10000, 30000, 0.05, 10)):
    X = sample(n, m, dens, 3)
    M = C.CountMatrix(X)
    iters = 200
    line = f"{name:28s} nnz {X.nnz:9d} rank {r:2d}:"
    for B in (4, 8):
        whs = [synth.random_state(n, m, r, hy, seed=b) for b in range(B)]
        rates = []
        for grid in (None, C.batch_grid(B)):
            engs = [C.VBEngine(M, r, grid=grid) for _ in range(B)]
            for eng, wh in zip(engs, whs):
                eng.set_state(wh["lw"], wh["lh"], wh["eh"])
            t0 = time.perf_counter()
            if grid is None:
                [e.run(hy, Itmax=iters, Tol=0.0) for e in engs]
            else:
                C.run_batch(engs, [hy] * B, Itmax=iters, Tol=0.0)
            rates.append(B * iters / (time.perf_counter() - t0))
            for e in engs:
                e.close()
        line += f"  B={B}: {rates[0]:8.0f} -> {rates[1]:8.0f} it/s (x {rates[1] / rates[0]:4.2f})"
    print(line, flush=True)
    M.close()
