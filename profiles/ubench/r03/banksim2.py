import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, scipy.sparse as sp
import ccfindr_amd as C
import util_layout as U
X = sp.load_npz('/tmp/c3.npz')
M = C.CountMatrix(X)
kGroupOf = np.array([0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0,1,1,1,1,0,0,0,0,0,0,0,0,1,1,1,1,
                     2,2,2,2,3,3,3,3,3,3,3,3,2,2,2,2,3,3,3,3,2,2,2,2,2,2,2,2,3,3,3,3])
def cyc_group(res):   # res [T,16] residues (all lanes read each step; idle -> residue 0)
    T = res.shape[0]
    cnt = np.zeros((T,16), int)
    for j in range(16): np.add.at(cnt, (np.arange(T), res[:, j]), 1)
    return cnt.max(1).sum(), T
def lane_major(locals_by_lane_phase, T, order_mode):
    # locals_by_lane_phase[j][ph] = array of residues; returns res [T,16]
    cnt = np.zeros((3,16,16), int)
    for j in range(16):
        for ph in range(3):
            for r in locals_by_lane_phase[j][ph]: cnt[ph,j,r] += 1
    rem = cnt.sum(2)   # [3,16]
    res = np.zeros((T,16), int)
    for t in range(T):
        used = np.zeros(16, int)
        cur = [0 if rem[0,j]>0 else (1 if rem[1,j]>0 else 2) for j in range(16)]
        if order_mode == 0: order = [(t + q) % 16 for q in range(16)]
        else:
            nopt = [int((cnt[cur[j], j] > 0).sum()) if rem[cur[j], j] > 0 else 99 for j in range(16)]
            order = sorted(range(16), key=lambda j: (nopt[j], (j - t) % 16))
        for j in order:
            ph = cur[j]
            if rem[ph, j] == 0: res[t, j] = 0; used[0] += 0; continue
            c = cnt[ph, j]
            cand = [r for r in range(16) if c[r] > 0 and used[r] == 0]
            if cand: r = max(cand, key=lambda r: (c[r], -r))
            else:
                cand = [r for r in range(16) if c[r] > 0]
                r = min(cand, key=lambda r: (used[r], -c[r]))
            res[t, j] = r; used[r] += 1; cnt[ph, j, r] -= 1; rem[ph, j] -= 1
    return res
rng = np.random.default_rng(0)
for side in (0, 1):
    v = U.build_layout(M, side, 10)
    rs = v["row_slots"]; packed = v["packed"]
    sl = rng.choice(v["n_slices"], 60, replace=False)
    tot = {"greedy":[0,0], "lm_rot":[0,0], "lm_mc":[0,0]}
    for s in sl:
        w = v["slice_width"][s]; off = v["slice_off"][s]
        blk = packed[off:off + w*64].reshape(w//4, 64, 4).transpose(0,2,1).reshape(w, 64)
        cnt = blk >> 18; local = ((blk & 0x3FFF0) >> 4) // rs
        for g in range(4):
            lanes = np.flatnonzero(kGroupOf == g)
            c, r = cyc_group((local[:, lanes] & 15)); tot["greedy"][0]+=c; tot["greedy"][1]+=r
            lbp = []
            for l in lanes:
                vv = cnt[:, l] > 0
                L = local[vv, l] & 15; Cn = cnt[vv, l]
                lbp.append([L[Cn==1], L[Cn==2], L[Cn>2]])
            for name, mode in (("lm_rot",0),("lm_mc",1)):
                res = lane_major(lbp, w, mode)
                c, r = cyc_group(res); tot[name][0]+=c; tot[name][1]+=r
    print("side", side, {k: round(a/b,3) for k,(a,b) in tot.items()})
