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
for _ in range(5): eng.step(bench.HYPER)
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
    W = T[side, :, 2:].reshape(nwg.value, nw.value, 2)
    dur = (W[:, :, 1] - W[:, :, 0]) / 100.0
    wend = (W[:, :, 1] - t0) / 100.0
    print(f"side {side}: wg start min/mean/max {ws.min():.1f} {ws.mean():.1f} {ws.max():.1f}; wg end min/mean/max {we.min():.1f} {we.mean():.1f} {we.max():.1f}; wg dur mean {np.mean(we-ws):.1f} max {np.max(we-ws):.1f} p90 {np.percentile(we-ws,90):.1f}")
    print(f"   wave end within wg: spread (max-min) mean {np.mean(wend.max(1)-wend.min(1)):.1f} max {np.max(wend.max(1)-wend.min(1)):.1f}")
    d = we - ws
    print("   wg duration histogram:", np.histogram(d, bins=8)[0].tolist(), np.round(np.histogram(d, bins=8)[1], 1).tolist())

# --- per-block view: does a workgroup's duration depend on which minor block it works on? ---
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from util_layout import build_layout
for side in (0, 1):
    v = build_layout(M, side, r)
    nwgv = v["n_wg"]
    wgmap = [(b % 8) * (nwgv // 8) + (b // 8) for b in range(nwgv)] if nwgv % 8 == 0 else list(range(nwgv))
    ws, we = (T[side, :, 0] - t0) / 100.0, (T[side, :, 1] - t0) / 100.0
    dur = np.zeros(nwgv)
    for b in range(nwgv):
        dur[wgmap[b]] = we[b] - ws[b]          # dbg rows are indexed by share id (wg), see the kernel
    dur = we - ws                                # T rows are already indexed by share id
    blk = np.array([v["seg_block"][v["wg_seg0"][g]] if v["wg_seg0"][g + 1] > v["wg_seg0"][g] else -1 for g in range(nwgv)])
    cost = np.array([(v["slice_width"][v["seg_ptr"][v["wg_seg0"][g]]:v["seg_ptr"][v["wg_seg0"][g + 1]]].astype(np.int64) + 10).sum() for g in range(nwgv)])
    nsl = np.array([v["seg_ptr"][v["wg_seg0"][g + 1]] - v["seg_ptr"][v["wg_seg0"][g]] for g in range(nwgv)])
    print(f"side {side}: corr(duration, model cost) = {np.corrcoef(dur, cost)[0,1]:.3f}; cost min/mean/max {cost.min()} {cost.mean():.0f} {cost.max()}; slices per wg {nsl.min()}..{nsl.max()}")
    for b in sorted(set(blk.tolist())):
        sel = blk == b
        print(f"   block {b:3d}: {sel.sum():2d} wgs, duration mean {dur[sel].mean():6.1f} min {dur[sel].min():6.1f} max {dur[sel].max():6.1f}; cost mean {cost[sel].mean():7.0f}")
