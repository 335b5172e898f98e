"""Soak of the batch path: 150 rounds of (create 12 engines of mixed ranks, one width; set states; VB batch run; ML batch run on 6 of
them; destroy), then the device's free memory and the process's RSS against the start."""
import os, sys, time, resource
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import ccfindr_amd as C
from ccfindr_amd import synth
X = synth.drop_empty(synth.simulate_data(600, (150, 200), seed=3, sparse=True))
M = C.CountMatrix(X)
n, m = X.shape
hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
ranks = [2, 3, 4, 5, 6, 7, 8, 2, 3, 4, 5, 6]
pad = C.engine.padded_rank(8)
grid = C.batch_grid(len(ranks))
rng = np.random.default_rng(0)
def one_round(k):
    engs = [C.VBEngine(M, r, grid=grid, pad_rank=pad) for r in ranks]
    for e, r in zip(engs, ranks):
        wh = synth.random_state(n, m, r, hy, seed=k * 100 + r)
        e.set_state(wh["lw"], wh["lh"], wh["eh"])
    out = C.run_batch(engs, [hy] * len(engs), Itmax=40, Tol=1e-4, n0=5)
    ml = [e for e, r in zip(engs, ranks) if r == 4 or r == 3][:4]
    same = [e for e in ml if e.rank == ml[0].rank]
    for e in same:
        e.ml_set_state(rng.uniform(0.1, 1, (n, e.rank)), rng.uniform(0.1, 1, (e.rank, m)))
    C.run_batch_ml(same, Itmax=25, Tol=0.0)
    for e in engs:
        e.close()
    return sum(o["it"] for o in out)
one_round(0)
torch.cuda.synchronize()
C.load().vbnmf_pool_trim()
free0 = torch.cuda.mem_get_info()[0]
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
t0 = time.perf_counter()
its = 0
marks = []
for k in range(1, 451):
    its += one_round(k)
    if k in (150, 450):                                    # (a leak grows with the rounds; a runtime's one-off bookkeeping does not)
        torch.cuda.synchronize()
        held = torch.cuda.mem_get_info()[0]
        C.load().vbnmf_pool_trim()                         # (freed device buffers are kept for reuse: give them back before looking)
        marks.append((k, held, torch.cuda.mem_get_info()[0], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss))
dt = time.perf_counter() - t0
print(f"450 rounds, {its} VB iterations, {dt:.2f} s; device memory free at the start {free0 / 2**20:.0f} MiB, peak RSS {rss0 / 1024:.0f} MiB")
for k, held, free, rss in marks:
    print(f"   after {k} rounds: free {free / 2**20:.0f} MiB ({held / 2**20:.0f} with the pool's buffers held), peak RSS {rss / 1024:.0f} MiB")
assert abs(marks[1][2] - marks[0][2]) < 32 << 20 and abs(free0 - marks[1][2]) < 256 << 20 and marks[1][3] - rss0 < 200 * 1024
print("soak ok")
