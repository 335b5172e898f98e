import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, ccfindr_amd as C
from ccfindr_amd import synth
HY = bench.HYPER
free0 = torch.cuda.mem_get_info()[0]
X = synth.drop_empty(synth.simulate_data(2000, (1000,) * 5, seed=1, sparse=True))
n, m = X.shape
for rep in range(60):
    M = C.CountMatrix(X)
    for r in (3, 10, 20):
        eng = C.VBEngine(M, r)
        wh = synth.random_state(n, m, r, HY, seed=rep)
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        eng.run(HY, Itmax=20, Tol=0.0)
        eng.ml_set_state(np.abs(wh["lw"]) + 0.1, np.abs(wh["lh"]) + 0.1)
        eng.ml_run(Itmax=10, Tol=0.0)
        eng.spmm(wh["lh"])
        eng.close()
    M.close()
torch.cuda.synchronize()
held = torch.cuda.mem_get_info()[0]
C.load().vbnmf_pool_trim()                     # (round 4: freed buffers wait in the library's pool for the next engine)
free1 = torch.cuda.mem_get_info()[0]
print(f"180 engines created and destroyed: device memory free before {free0 / 2**20:.0f} MiB, after {free1 / 2**20:.0f} MiB "
      f"({(free1 - held) / 2**20:.0f} MiB were held by the buffer pool)", flush=True)
assert abs(free0 - free1) < 256 * 2**20
name, X, r = bench.make_workload(False)
n, m = X.shape
eng = C.VBEngine(C.CountMatrix(X), r)
wh = synth.random_state(n, m, r, HY, seed=1003)
eng.set_state(wh["lw"], wh["lh"], wh["eh"])
t0 = time.perf_counter()
out = eng.run(HY, Itmax=20000, Tol=0.0, flags=(True,) * 4)
dt = time.perf_counter() - t0
print(f"C3 soak: {out['it']} steps with hyper updates in {dt:.2f} s ({out['it'] / dt:.0f} it/s), reason {out['reason']}, lkh {out['lkh']:.12g}, hyper {out['hyper']}", flush=True)
a = eng.get_state(("ew",))["ew"]
assert np.isfinite(a).all() and np.isfinite(out["lkh"])
eng.close()
print("soak ok")

# round 2: the partitioned device-driven loop (local group of 4 partitions) and the device-side initialisers, repeatedly
X = synth.drop_empty(synth.simulate_data(1500, (700,) * 4, seed=2, sparse=True))
n, m = X.shape
M = C.CountMatrix(X)
from ccfindr_amd.parallel import cell_partition
for rep in range(20):
    cuts = cell_partition(m, 4)
    comm = C.Communicator.local(4)
    parts = [C.VBEngine(M, 8, cols=c, m_global=m) for c in cuts]
    for p in parts:
        p.attach_comm(comm)
        p.random_state(HY, seed=rep)
    comm.state_finish()
    res = comm.run(HY, Itmax=300, Tol=1e-6, flags=(True,) * 4)
    assert np.isfinite(res["lkh"]) and res["it"] >= 11
    for p in parts:
        p.close()
    comm.close()
    eng = C.VBEngine(M, 14)
    u, d, vt, it = eng.svd(4)
    assert np.isfinite(d).all()
    eng.close()
torch.cuda.synchronize()
C.load().vbnmf_pool_trim()
free2 = torch.cuda.mem_get_info()[0]
print(f"20 x (4-partition group loop + device SVD): device memory free {free2 / 2**20:.0f} MiB", flush=True)
assert abs(free2 - free1) < 1024 * 2**20
print("soak 2 ok")
