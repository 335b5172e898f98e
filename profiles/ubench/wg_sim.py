"""CPU: simulate the in-workgroup ticket scheduling of the C3 layout: per segment, waves pull slices (in list order) -> makespan vs ideal."""
import sys, heapq, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import bench, ccfindr_amd as C
from util_layout import build_layout
name, X, r = bench.make_workload(False)
M = C.CountMatrix(X)
for side in (0, 1):
    v = build_layout(M, side, r)
    sw = np.asarray(v["slice_width"]); sf = np.asarray(v["slice_fast"])
    seg_ptr = np.asarray(v["seg_ptr"]); wg_seg0 = np.asarray(v["wg_seg0"])
    nw = 12
    disc = 0.4 if side == 0 else 0.1
    tot_ideal = []; tot_make = []; nsl = []
    for wg in range(v["n_wg"]):
        ideal = 0.0; make = 0.0; cnt = 0
        for seg in range(wg_seg0[wg], wg_seg0[wg + 1]):
            cost = [sw[s] - disc * sf[s] + 10 for s in range(seg_ptr[seg], seg_ptr[seg + 1])]
            cnt += len(cost)
            h = [0.0] * nw; heapq.heapify(h)
            for c in cost:
                t = heapq.heappop(h); heapq.heappush(h, t + c)
            make += max(h); ideal += sum(cost) / nw
        tot_ideal.append(ideal); tot_make.append(make); nsl.append(cnt)
    ti, tm = np.array(tot_ideal), np.array(tot_make)
    print(f"side {side}: slices/wg {min(nsl)}..{max(nsl)}; widths min/mean/max {sw.min()} {sw.mean():.1f} {sw.max()}; ideal per-wave cost mean {ti.mean():.1f}; makespan mean {tm.mean():.1f} max {tm.max():.1f}; makespan/ideal mean {np.mean(tm/ti):.3f} max {np.max(tm/ti):.3f}; max makespan / mean ideal {tm.max()/ti.mean():.3f}")
    # lane utilisation: sum of task lengths / (64 * slice width)
    tl = np.asarray(v["task_len"]) if "task_len" in v else None
    if tl is not None:
        print("   lane utilisation", tl.sum() / (64.0 * sw.sum()))
