import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, bench, ccfindr_amd as C
from util_layout import build_layout
name, X, r = bench.make_workload(False)
M = C.CountMatrix(X)
for side in (0, 1):
    v = build_layout(M, side, r)
    pk = v["packed"]; cnt = pk >> 18
    sw = v["slice_width"]; so = v["slice_off"]; sf = v["slice_fast"]
    tot = np.zeros(6)   # [fast stretch slots, general: x==0(pad), x==1, x==2, x==3..4, x>=5]
    for s in range(0, v["n_slices"], 7):     # sample every 7th slice
        w, o, f = int(sw[s]), int(so[s]), int(sf[s])
        c = cnt[o:o + w * 64].reshape(w // 4, 64, 4).transpose(0, 2, 1).reshape(w, 64)   # [t][lane]
        tot[0] += f * 64
        g = c[f:]
        tot[1] += (g == 0).sum(); tot[2] += (g == 1).sum(); tot[3] += (g == 2).sum(); tot[4] += ((g >= 3) & (g <= 4)).sum(); tot[5] += (g >= 5).sum()
        # how long a stretch "all lanes <= 2" would be beyond f (entries sorted ones first; others in minor order?)
    print("side", side, "fractions of slots: fast %.3f | general: pad %.3f ones %.3f twos %.3f 3-4 %.3f >=5 %.3f" % tuple(tot / tot.sum()))
