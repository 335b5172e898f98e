#!/usr/bin/env python3
"""profiles/ubench/r05/update2_stamps_small.py -- k_update2's in-kernel time line on C1 (200 x 500, rank 3) and C2 (2 000 x 10 000
dense, rank 5 -- needs R = 6 in the instrumented build; skipped when absent).  Library built with -DVBNMF_ABL_STAMPS (VBNMF_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["VBNMF_UPDATE_PAIR"] = "1"
import numpy as np, ccfindr_amd as C
from ccfindr_amd import synth
HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
order = [(0, 1, "entry -> loads of the block's table row issued"), (1, 8, "fold: loads + column sums of the three tables"),
         (8, 9, "fold: block sum of the evidence partials"), (9, 10, "fold: evidence, control block written"), (10, 2, "fold: closing barrier"),
         (2, 3, "lga, the two rates"), (3, 11, "gene-side visits (thread 0)"), (11, 4, "cell-side visits (thread 0)"),
         (4, 5, "barrier (all threads done)"), (5, 6, "reduction tree (6 sums)"), (6, 7, "block partials written")]
for name, X, r in (("C1 200 x 500 rank 3", synth.drop_empty(synth.simulate_data(200, (100, 150, 250), seed=1, sparse=False)), 3),
                   ("1030 x 450 rank 4", synth.fill_empty(synth.simulate_data(1030, [150] * 3, alpha0=0.3, seed=4, depth=np.full(450, 900))), 4),
                   ("5000 x 20000 rank 10", synth.fill_empty(synth.simulate_data(5000, [4000] * 5, alpha0=0.1, seed=3, depth=np.full(20000, 400))), 10)):
    n, m = X.shape
    eng = C.VBEngine(C.CountMatrix(X), r)
    wh = synth.random_state(n, m, r, HY, seed=1003)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    eng.run(HY, Itmax=300, Tol=0.0, flags=(False,) * 4)
    lib = C.load()
    buf = (ctypes.c_ulonglong * (2 * 256 * 12))()
    lib.vbnmf_test_update_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    assert lib.vbnmf_test_update_stamps(buf) == 0
    v = np.frombuffer(buf, dtype=np.uint64).reshape(2, 256, 12).astype(np.int64)[0]
    t0 = v[:, 0].min()
    print(f"{name}: k_update2 blocks enter within {(v[:, 0].max() - t0) / 100:.2f} us and end between {(v[:, 7].min() - t0) / 100:.2f} and {(v[:, 7].max() - t0) / 100:.2f} us")
    for a, b, what in order:
        d = (v[:, b] - v[:, a]) / 100.0
        print(f"   {what:52s} mean {d.mean():6.2f} us  (min {d.min():5.2f}, max {d.max():5.2f})")
    eng.close()
