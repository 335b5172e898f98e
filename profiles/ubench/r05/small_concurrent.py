"""Small matrices (the reference's shipped data set is 1030 x 450): a step is latency bound (~32 us, two or three dependent kernels),
so the rank sweep's independent (run, rank) units are the parallelism there is.  vb_factorize(concurrent=K) keeps K units in flight
on K streams from K host threads: aggregate iterations per second against K, fixed number of iterations per unit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import ccfindr_amd as C
from ccfindr_amd import synth

X = synth.drop_empty(synth.simulate_data(1030, (150, 150, 150), seed=3, sparse=True))
M = C.CountMatrix(X)
ranks = list(range(2, 10))
nrun = 4
Itmax = 600
kw = dict(ranks=ranks, nrun=nrun, verbose=0, Tol=0.0, seed=5, Itmax=Itmax, unif_stop=False, hyper_update_n0=10)
units = len(ranks) * nrun
C.vb_factorize(M, ranks=[2, 3], nrun=1, verbose=0, Tol=0.0, seed=1, Itmax=20)        # warm the device and the layouts
print(f"{X.shape[0]} x {X.shape[1]}, nnz {X.nnz}: {units} units (ranks {ranks[0]}..{ranks[-1]} x {nrun} runs) of {Itmax} iterations each")
base = None
for K in (1, 2, 4, 8, 16):
    t0 = time.perf_counter()
    out = C.vb_factorize(M, concurrent=K, batch=1, **kw) if K > 1 else C.vb_factorize(M, batch=1, **kw)
    dt = time.perf_counter() - t0
    its = units * Itmax / dt
    base = base or its
    print(f"   concurrent={K:2d}: {dt:6.3f} s  {its:9.0f} iterations/s in all  ({its / base:4.2f} x)", flush=True)
# the restarts of a rank stepped by ONE launch (vbnmf_batch_run): nrun restarts per rank, batches of B
for nrun_b, B in ((4, 2), (4, 4), (8, 8), (16, 16), (32, 32)):
    kwb = dict(kw, nrun=nrun_b)
    t0 = time.perf_counter()
    out = C.vb_factorize(M, batch=B, **kwb)
    dt = time.perf_counter() - t0
    its = len(ranks) * nrun_b * Itmax / dt
    print(f"   batch={B:2d} ({nrun_b:2d} runs per rank): {dt:6.3f} s  {its:9.0f} iterations/s in all  ({its / base:4.2f} x)", flush=True)
M.close()
