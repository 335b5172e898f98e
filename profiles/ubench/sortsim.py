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
    lens = []; n1s = []; n2s = []; blks = []
    for s in range(v["n_slices"]):
        w, o = int(sw[s]), int(so[s])
        c = cnt[o:o + w * 64].reshape(w // 4, 64, 4).transpose(0, 2, 1).reshape(w, 64)
        lens.append((c > 0).sum(axis=0)); n1s.append((c == 1).sum(axis=0)); n2s.append((c == 2).sum(axis=0)); blks.append(np.full(64, sb[s]))
    L = np.concatenate(lens); N1 = np.concatenate(n1s); N2 = np.concatenate(n2s); B = np.concatenate(blks)
    keep = L > 0
    L, N1, N2, B = L[keep], N1[keep], N2[keep], B[keep]
    fc, gc = COST[side]
    def simulate(qsort, qwidth, label, two=False):
        slots = fast = fast2 = 0
        for b in np.unique(B):
            idx = np.where(B == b)[0]
            key1 = -((L[idx] + qsort - 1) // qsort)
            order = idx[np.lexsort((-N1[idx], key1))]
            for i in range(0, len(order), 64):
                t = order[i:i + 64]
                w = (L[t].max() + qwidth - 1) // qwidth * qwidth
                f = N1[t].min() // 8 * 8 if len(t) == 64 else 0
                f2 = max(f, (N1[t] + N2[t]).min() // 8 * 8) if len(t) == 64 else f
                slots += w; fast += min(f, w); fast2 += min(f2, w)
        cost = fast * fc + (slots - fast) * gc
        extra = ""
        if two:
            c2 = fast * fc + (fast2 - fast) * (fc + 8.5) + (slots - fast2) * gc
            extra = "  with a <=2 stretch: fast2 %.3f, cost/nnz %.2f" % (fast2 / slots, c2 / (M.nnz / 64))
        print("side %d %-28s slots/nnz %.4f fast %.3f cost/nnz %.2f%s" % (side, label, slots * 64 / M.nnz, fast / slots, cost / (M.nnz / 64), extra))
    simulate(8, 8, "sort 8, width 8 (before)")
    simulate(4, 4, "sort 4, width 4 (now)", two=True)
    simulate(8, 4, "sort 8, width 4")
    simulate(16, 4, "sort 16, width 4")
    simulate(32, 4, "sort 32, width 4", two=True)
