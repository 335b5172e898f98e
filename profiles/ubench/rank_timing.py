import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, ccfindr_amd as C
from ccfindr_amd import synth
name, X, _ = bench.make_workload(False)
n, m = X.shape
M = C.CountMatrix(X)
for r in [int(x) for x in os.environ.get("RANKS", "10,12,14,16,18,20").split(",")]:
    eng = C.VBEngine(M, r)
    wh = synth.random_state(n, m, r, bench.HYPER, seed=r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    for _ in range(5): eng.step(bench.HYPER)
    eng.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(40): lkh, _ = eng.step(bench.HYPER)
    dt = (time.perf_counter() - t0) / 40
    ms, cnt = eng.timing_get()
    print(f"rank {r:2d}: step {dt*1e6:7.1f} us  sweep {ms/cnt*1e3:7.1f} us  it/s {1/dt:7.1f}  lkh {lkh:.6f}", flush=True)
    eng.close()
