"""factorize()'s restarts (reference R/factorize.R:181, nrun defaults to 20) one loop at a time against one launch for all (vbnmf_batch_ml_run):
the 1030 x 450 sample and config C2's size, aggregate ML-NMF iterations per second."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import ccfindr_amd as C
from ccfindr_amd import synth
import scipy.sparse as sp

def dense_sample(n, m, density, seed):
    rng = np.random.default_rng(seed)
    X = sp.random(n, m, density=density, format="csc", random_state=rng, data_rvs=lambda k: rng.integers(1, 6, k).astype(np.float64))
    X = X + sp.csc_matrix((np.ones(n), (np.arange(n), rng.integers(0, m, n))), shape=(n, m)) + sp.csc_matrix((np.ones(m), (rng.integers(0, n, m), np.arange(m))), shape=(n, m))
    return sp.csc_matrix(X)

for name, X, r in (("1030 x 450 sample", synth.drop_empty(synth.simulate_data(1030, (150, 150, 150), seed=3, sparse=True)), 5),
                   ("2 000 x 10 000, 75 % (C2's size)", dense_sample(2000, 10000, 0.75, 3), 5)):
    M = C.CountMatrix(X)
    n, m = X.shape
    iters = 300
    rng = np.random.default_rng(1)
    line = f"{name:34s} nnz {X.nnz:9d} rank {r}:"
    for B in (4, 20):
        starts = [(rng.uniform(0.1, 1, (n, r)), rng.uniform(0.1, 1, (r, m))) for _ in range(B)]
        rates = []
        for grid in (None, C.batch_grid(B)):
            engs = [C.VBEngine(M, r, grid=grid) for _ in range(B)]
            for e, (w, h) in zip(engs, starts):
                e.ml_set_state(w, h)
            t0 = time.perf_counter()
            if grid is None:
                [e.ml_run(Itmax=iters, Tol=0.0) for e in engs]
            else:
                C.run_batch_ml(engs, Itmax=iters, Tol=0.0)
            rates.append(B * iters / (time.perf_counter() - t0))
            for e in engs:
                e.close()
        line += f"  B={B}: {rates[0]:8.0f} -> {rates[1]:8.0f} it/s (x {rates[1] / rates[0]:4.2f})"
    print(line, flush=True)
    M.close()
# through factorize(): ranks 2..5, nrun = 20 (the reference's default), criterion 'likelihood'
X = synth.drop_empty(synth.simulate_data(1030, (150, 150, 150), seed=3, sparse=True))
for batch in (1, None):
    t0 = time.perf_counter()
    res = C.factorize(X, ranks=[2, 3, 4, 5], nrun=20, verbose=0, Tol=1e-5, Itmax=2000, seed=7, batch=batch)
    dt = time.perf_counter() - t0
    its = sum(sum(s) for s in res.nsteps)
    print(f"factorize(ranks 2..5, nrun = 20) batch={batch}: {dt:6.2f} s, {its} iterations, {its / dt:8.0f} it/s", flush=True)
