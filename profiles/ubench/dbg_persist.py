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
def snap():
    buf = np.zeros(cnt, dtype=np.uint64)
    N.check(L.vbnmf_engine_debug_times(eng._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cnt, None, None))
    T = buf.reshape(2, nwg.value, rec).astype(np.int64)
    d0 = (T[0, :, 1] - T[0, :, 0]) / 100.0
    d1 = (T[1, :, 1] - T[1, :, 0]) / 100.0
    end = (T[1, :, 1] - T[0, :, 0].min()) / 100.0
    return d0, d1, end
for _ in range(400): eng.step(bench.HYPER)
S = []
for k in range(6):
    for _ in range(37): eng.step(bench.HYPER)
    S.append(snap())
D0 = np.array([s[0] for s in S]); D1 = np.array([s[1] for s in S]); E = np.array([s[2] for s in S])
print("gene side: mean dur %.1f, per-wg std across snapshots %.2f, std of per-wg means %.2f" % (D0.mean(), D0.std(axis=0).mean(), D0.mean(axis=0).std()))
print("cell side: mean dur %.1f, per-wg std across snapshots %.2f, std of per-wg means %.2f" % (D1.mean(), D1.std(axis=0).mean(), D1.mean(axis=0).std()))
print("kernel end: mean over snapshots of (max - mean) %.1f; of the per-wg MEAN end: max - mean %.1f" % ((E.max(axis=1) - E.mean(axis=1)).mean(), E.mean(axis=0).max() - E.mean()))
c = np.corrcoef(D0[0] + D1[0], D0[3] + D1[3])[0, 1]
print("corr of per-wg total between snapshot 0 and 3: %.2f; corr(gene, cell) within snapshot: %.2f" % (c, np.corrcoef(D0[0], D1[0])[0, 1]))
mean_tot = (D0 + D1).mean(axis=0)
order = np.argsort(mean_tot)
print("slowest 8 wgs (index: mean total):", [(int(i), round(float(mean_tot[i]), 1)) for i in order[-8:]])
print("fastest 8 wgs:", [(int(i), round(float(mean_tot[i]), 1)) for i in order[:8]])
print("by XCD slot (wg // 32): ", np.round([mean_tot[i * 32:(i + 1) * 32].mean() for i in range(8)], 1))
