"""nrun = 1 is the reference's default: a rank sweep 2..9 of the 1030 x 450 sample, one loop at a time against ONE batch over the
ranks (engines made as wide as rank 9's, vbnmf_set_engine_padding) -- and the same with five restarts per rank."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import ccfindr_amd as C
from ccfindr_amd import synth
X = synth.drop_empty(synth.simulate_data(1030, (150, 150, 150), seed=3, sparse=True))
M = C.CountMatrix(X)
C.vb_factorize(M, ranks=[2, 3], nrun=1, verbose=0, Tol=0.0, seed=1, Itmax=20)
for nrun in (1, 5):
    kw = dict(ranks=range(2, 10), nrun=nrun, verbose=0, Tol=0.0, seed=5, Itmax=600, unif_stop=False, hyper_update_n0=10)
    units = 8 * nrun
    for label, extra in (("one loop at a time", dict(batch=1)), ("restarts of a rank together", dict(across_ranks=False)), ("default (across ranks)", {})):
        if nrun == 1 and label.startswith("restarts"):
            continue
        t0 = time.perf_counter()
        C.vb_factorize(M, **kw, **extra)
        dt = time.perf_counter() - t0
        print(f"nrun {nrun}: {label:28s} {dt:6.3f} s  {units * 600 / dt:9.0f} iterations/s in all", flush=True)
M.close()
