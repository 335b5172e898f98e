#!/usr/bin/env python3
"""profiles/ubench/c2_prof.py -- BASELINE config C2 (2 000 x 10 000 dense counts, ~75 % non-zero, rank 5) through the
device-driven loop, for `rocprofv3 --kernel-trace --stats`: where the step time of a small dense matrix goes.
Prints steps/s and the layout facts the close-out of the MFMA row uses."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ccfindr_amd as C          # noqa: E402
from ccfindr_amd import synth    # noqa: E402

HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
X = synth.fill_empty(synth.simulate_data(2000, [2000] * 5, alpha0=2.0, seed=2, depth=np.full(10000, 4000)), seed=2)
n, m = X.shape
eng = C.VBEngine(C.CountMatrix(X), 5)
wh = synth.random_state(n, m, 5, HY, seed=1002)
eng.set_state(wh["lw"], wh["lh"], wh["eh"])
for _ in range(10):
    eng.step(HY)
eng.timing_enable(True)
for _ in range(50):
    eng.step(HY)
ms, cnt = eng.timing_get()
eng.timing_enable(False)
t0 = time.perf_counter()
res = eng.run(HY, Itmax=steps, Tol=0.0, flags=(False,) * 4)
dt = time.perf_counter() - t0
info = eng.layout_info()
print(f"C2 {n} x {m}, nnz {X.nnz} ({X.nnz / n / m:.3f}), rank 5: {res['it'] / dt:.0f} steps/s device-driven = {1e6 * dt / res['it']:.1f} us/step; "
      f"k_sweep {1e3 * ms / max(cnt, 1):.1f} us by HIP events (host-stepped); slots gene/cell {info['slots_gene_side']}/{info['slots_cell_side']}, "
      f"tasks {info['tasks_gene_side']}/{info['tasks_cell_side']}")
eng.close()
