"""Which grids should the engines of a batch sit on?  Sixteen (and eight, thirty-two) engines of rank 5 on the 1030 x 450 sample, 400
iterations, stepped by one launch: aggregate iterations per second against (sweep workgroups, update blocks) per engine."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import ccfindr_amd as C
from ccfindr_amd import synth
X = synth.drop_empty(synth.simulate_data(1030, (150, 150, 150), seed=3, sparse=True))
M = C.CountMatrix(X)
n, m = X.shape
r, iters = 5, 400
hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
SMALLER = len(sys.argv) > 1 and sys.argv[1] == "smaller"
for B in ((2, 4, 8, 16) if SMALLER else (8, 16, 32)):
    whs = [synth.random_state(n, m, r, hy, seed=b) for b in range(B)]
    g = max(8, 256 // B)
    for grid in (((g, g), (g // 2, g // 2), (g // 4, g // 4), (g // 2, g), (g, g // 2)) if SMALLER else
                 ((g, g), (2 * g, g), (4 * g, g), (g, 2 * g), (2 * g, 2 * g), (8 * g if 8 * g <= 256 else 256, g))):
        if min(grid) < 8:
            continue
        best = 0.0
        for rep in range(3):
            engs = [C.VBEngine(M, r, grid=grid) for _ in range(B)]
            for eng, wh in zip(engs, whs):
                eng.set_state(wh["lw"], wh["lh"], wh["eh"])
            t0 = time.perf_counter()
            res = C.run_batch(engs, [hy] * B, Itmax=iters, Tol=0.0)
            dt = time.perf_counter() - t0
            best = max(best, B * iters / dt)
            for e in engs:
                e.close()
        print(f"B {B:2d}  sweep workgroups {grid[0]:3d}  update blocks {grid[1]:3d}: {best:9.0f} iterations/s in all", flush=True)
