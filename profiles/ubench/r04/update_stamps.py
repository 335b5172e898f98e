#!/usr/bin/env python3
"""profiles/ubench/r04/update_stamps.py [rank] -- where k_update's microseconds go.  Needs a library built with
-DVBNMF_ABL_STAMPS (VBNMF_LIB): thread 0 of every block stamps wall_clock64() (100 MHz) at eight points of the kernel; the
stamps of the LAST step of a device-driven run on the headline matrix are read back and reported per side as the mean over
the 256 blocks of every interval, plus the spread of the blocks' start and end."""
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
s = np.frombuffer(buf, dtype=np.uint64).reshape(2, 256, 12).astype(np.int64)
names = ["", "entry -> inverse-index stretch known, staging loads issued", "-> prologue done (control fold / column sums, barrier)", "-> lga = f(psi, lgamma, log of a)",
         "-> main loop done (thread 0)", "-> barrier (all threads done)", "-> reduction tree done", "-> block partials written"]
for side, label in ((0, "W (gene side, with the control fold)"), (1, "H (cell side)")):
    v = s[side]
    t0 = v[:, 0].min()
    print(f"rank {rank} k_update {label}: first block enters at 0, last at {(v[:, 0].max() - t0) / 100:.2f} us; "
          f"blocks end between {(v[:, 7].min() - t0) / 100:.2f} and {(v[:, 7].max() - t0) / 100:.2f} us")
    for i in range(1, 8):
        d = (v[:, i] - v[:, i - 1]) / 100.0
        print(f"   {names[i]:58s} mean {d.mean():6.2f} us  (min {d.min():5.2f}, max {d.max():5.2f})")
    if side == 0:                                   # inside the control fold: stamps 8 (column sums done), 9 (evidence partials summed), 10 (hyper_update done)
        for a, b, what in ((1, 8, "fold: loads + column sums of the two partial tables"), (8, 9, "fold: block sum of the evidence partials (2 barriers)"),
                           (9, 10, "fold: evidence, Newton recurrences, control block written"), (10, 2, "fold: closing barrier")):
            d = (v[:, b] - v[:, a]) / 100.0
            print(f"      {what:55s} mean {d.mean():6.2f} us  (min {d.min():5.2f}, max {d.max():5.2f})")
    d = (v[:, 7] - v[:, 0]) / 100.0
    print(f"   {'block lifetime':58s} mean {d.mean():6.2f} us  (min {d.min():5.2f}, max {d.max():5.2f})")
