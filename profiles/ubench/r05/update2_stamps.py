#!/usr/bin/env python3
"""profiles/ubench/r05/update2_stamps.py [rank] -- where k_update2's microseconds go.  Needs a library built with
-DVBNMF_ABL_STAMPS (VBNMF_LIB): thread 0 of every block stamps wall_clock64() (100 MHz) at twelve points of the kernel; the
stamps of the LAST step of a device-driven run on the headline matrix are read back and reported as the mean over the 256
blocks of every interval, plus the spread of the blocks' start and end."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench, ccfindr_amd as C
from ccfindr_amd import synth

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 10
name, X, _ = bench.make_workload(False)
n, m = X.shape
eng = C.VBEngine(C.CountMatrix(X), rank)
wh = synth.random_state(n, m, rank, bench.HYPER, seed=1003)
eng.set_state(wh["lw"], wh["lh"], wh["eh"])
eng.run(bench.HYPER, Itmax=300, Tol=0.0, flags=(True,) * 4)
lib = C.load()
buf = (ctypes.c_ulonglong * (2 * 256 * 12))()
lib.vbnmf_test_update_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert lib.vbnmf_test_update_stamps(buf) == 0
v = np.frombuffer(buf, dtype=np.uint64).reshape(2, 256, 12).astype(np.int64)[0]
order = [(0, 1, "entry -> loads of the block's table row issued"),
         (1, 8, "fold: loads + column sums of the three tables (table row to LDS behind them)"),
         (8, 9, "fold: block sum of the evidence partials (2 barriers)"),
         (9, 10, "fold: evidence, Newton recurrences, control block written"),
         (10, 2, "fold: closing barrier"),
         (2, 3, "lga (psi, lgamma, log of aw and ah), the two rates"),
         (3, 11, "gene-side stretch (thread 0: gather, posterior, stores)"),
         (11, 4, "cell-side stretch (thread 0)"),
         (4, 5, "barrier (all threads done)"),
         (5, 6, "reduction tree (6 sums)"),
         (6, 7, "block partials written")]
t0 = v[:, 0].min()
print(f"rank {rank} k_update2: first block enters at 0, last at {(v[:, 0].max() - t0) / 100:.2f} us; "
      f"blocks end between {(v[:, 7].min() - t0) / 100:.2f} and {(v[:, 7].max() - t0) / 100:.2f} us")
for a, b, what in order:
    d = (v[:, b] - v[:, a]) / 100.0
    print(f"   {what:62s} mean {d.mean():6.2f} us  (min {d.min():5.2f}, max {d.max():5.2f})")
d = (v[:, 7] - v[:, 0]) / 100.0
print(f"   {'block lifetime':62s} mean {d.mean():6.2f} us  (min {d.min():5.2f}, max {d.max():5.2f})")
