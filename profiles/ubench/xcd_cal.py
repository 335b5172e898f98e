"""Measure per-XCD workgroup durations, derive speed weights, print them as VBNMF_XCD_WEIGHTS (no trailing newline junk)."""
import os, sys, ctypes, numpy as np
os.environ["VBNMF_DEBUG_TIMES"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, ccfindr_amd as C
from ccfindr_amd import synth, _native as N
name, X, r = bench.make_workload(False)
n, m = X.shape
M = C.CountMatrix(X); eng = C.VBEngine(M, r)
wh = synth.random_state(n, m, r, bench.HYPER, seed=1003)
eng.set_state(wh["lw"], wh["lh"], wh["eh"])
L = N.load(); nwg = ctypes.c_int32(); nw = ctypes.c_int32()
N.check(L.vbnmf_engine_debug_times(eng._h, None, 0, ctypes.byref(nwg), ctypes.byref(nw)))
rec = 2 + 2 * nw.value
cnt = 2 * nwg.value * rec
tot = np.zeros(nwg.value)
for _ in range(1500): eng.step(bench.HYPER)
K = 10
for k in range(K):
    for _ in range(23): eng.step(bench.HYPER)
    buf = np.zeros(cnt, dtype=np.uint64)
    N.check(L.vbnmf_engine_debug_times(eng._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cnt, None, None))
    T = buf.reshape(2, nwg.value, rec).astype(np.int64)
    tot += ((T[0, :, 1] - T[0, :, 0]) + (T[1, :, 1] - T[1, :, 0])) / 100.0
tot /= K
x = np.array([tot[i * 32:(i + 1) * 32].mean() for i in range(8)])
w = x.mean() / x
sys.stderr.write("per-XCD mean total us: %s ; kernel-end excess (max-mean) %.1f\n" % (np.round(x, 1), tot.max() - tot.mean()))
print(",".join("%.4f" % v for v in w))
