import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, bench, ccfindr_amd as C
from util_layout import build_layout
name, X, r = bench.make_workload(False)
M = C.CountMatrix(X)
COST = {0: (27.5, 44.0), 1: (25.3, 28.3)}
for side in (0, 1):
    v = build_layout(M, side, r)
    cnt = (v["packed"] >> 18)
    sw = v["slice_width"]; so = v["slice_off"]; sb = v["slice_block"]
    lens = []; n1s = []; blks = []
    for s in range(v["n_slices"]):
        w, o = int(sw[s]), int(so[s])
        c = cnt[o:o + w * 64].reshape(w // 4, 64, 4).transpose(0, 2, 1).reshape(w, 64)
        lens.append((c > 0).sum(axis=0)); n1s.append((c == 1).sum(axis=0)); blks.append(np.full(64, sb[s]))
    L = np.concatenate(lens); N1 = np.concatenate(n1s); B = np.concatenate(blks)
    keep = L > 0
    L, N1, B = L[keep], N1[keep], B[keep]
    fc, gc = COST[side]
    def simulate(qsort, qwidth, label, zigzag=False, fq=8):
        slots = fast = 0
        for b in np.unique(B):
            idx = np.where(B == b)[0]
            cls = (L[idx] + qsort - 1) // qsort
            k2 = -N1[idx].astype(np.int64)
            if zigzag: k2 = np.where(cls % 2 == 0, k2, -k2)
            order = idx[np.lexsort((k2, -cls))]
            for i in range(0, len(order), 64):
                t = order[i:i + 64]
                w = (L[t].max() + qwidth - 1) // qwidth * qwidth
                f = N1[t].min() // fq * fq if len(t) == 64 else 0
                slots += w; fast += min(f, w)
        cost = fast * fc + (slots - fast) * gc
        print("side %d %-34s slots/nnz %.4f fast %.3f cost/nnz %.2f" % (side, label, slots * 64 / M.nnz, fast / slots, cost / (M.nnz / 64)))
    simulate(4, 4, "sort 4, width 4 (now)")
    simulate(4, 4, "sort 4, width 4, zigzag", zigzag=True)
    simulate(8, 4, "sort 8, width 4, zigzag", zigzag=True)
    simulate(4, 4, "sort 4, zigzag, stretch quantum 4", zigzag=True, fq=4)
