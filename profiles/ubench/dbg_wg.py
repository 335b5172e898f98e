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
for _ in range(300): eng.step(bench.HYPER)
L = N.load(); nwg = ctypes.c_int32(); nw = ctypes.c_int32()
N.check(L.vbnmf_engine_debug_times(eng._h, None, 0, ctypes.byref(nwg), ctypes.byref(nw)))
rec = 2 + 2 * nw.value
cnt = 2 * nwg.value * rec
buf = np.zeros(cnt, dtype=np.uint64)
N.check(L.vbnmf_engine_debug_times(eng._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cnt, None, None))
T = buf.reshape(2, nwg.value, rec).astype(np.int64)
t0 = T[0, :, 0].min()
for side in (0, 1):
    ws, we = (T[side, :, 0] - t0) / 100.0, (T[side, :, 1] - t0) / 100.0
    W = (T[side, :, 2:].reshape(nwg.value, nw.value, 2) - t0) / 100.0
    print(f"side {side}: wg start min/mean/max {ws.min():.1f} {ws.mean():.1f} {ws.max():.1f}; wg end {we.min():.1f} {we.mean():.1f} {we.max():.1f}; dur mean {np.mean(we-ws):.1f} max {np.max(we-ws):.1f}")
    print("   wave end - wave start (mean over wgs) by wave:", np.round((W[:, :, 1] - W[:, :, 0]).mean(axis=0), 1))
    d = we - ws
    print("   duration by groups of 32 wgs (layout order):", np.round([d[i:i + 32].mean() for i in range(0, nwg.value, 32)], 1))
