import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, scipy.sparse as sp
import ccfindr_amd as C
import util_layout as U
X = sp.load_npz('/tmp/c3.npz')
M = C.CountMatrix(X)
kGroupOf = np.array([0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0,1,1,1,1,0,0,0,0,0,0,0,0,1,1,1,1,
                     2,2,2,2,3,3,3,3,3,3,3,3,2,2,2,2,3,3,3,3,2,2,2,2,2,2,2,2,3,3,3,3])
def cycles(res, valid):
    # res: [T, 64] residues, valid: [T,64] bool ; returns total cycles and group-reads
    tot = 0; reads = 0
    for g in range(4):
        lanes = np.flatnonzero(kGroupOf == g)
        r = res[:, lanes]; v = valid[:, lanes]
        cnt = np.zeros((r.shape[0], 16), int)
        for j in range(16):
            np.add.at(cnt, (np.arange(r.shape[0])[v[:, j]], r[v[:, j], j]), 1)
        mx = cnt.max(1)
        tot += np.maximum(mx, 1).sum(); reads += r.shape[0]
    return tot, reads
rng = np.random.default_rng(0)
for side in (0, 1):
    v = U.build_layout(M, side, 10)
    rs = v["row_slots"]; packed = v["packed"]
    sl = rng.choice(v["n_slices"], 150, replace=False)
    res_cur = [0,0]; res_loc = [0,0]; res_loc2=[0,0]; res_rand=[0,0]
    for s in sl:
        w = v["slice_width"][s]; off = v["slice_off"][s]
        blk = packed[off:off + w*64].reshape(w//4, 64, 4).transpose(0,2,1).reshape(w, 64)   # [t, lane]
        cnt = blk >> 18
        local = ((blk & 0x3FFF0) >> 4) // rs
        valid = cnt > 0
        # padded slots: lane reads row 0 -> still an LDS read; count them as valid reads of residue 0? they are reads. keep all.
        allv = np.ones_like(valid)
        c, r = cycles(local & 15, allv); res_cur[0]+=c; res_cur[1]+=r
        # lane-local: per lane, phases (1, 2, other), within phase cyclic residue starting at lane pos in group
        new = np.zeros_like(local); 
        pos_in_group = np.zeros(64, int)
        for g in range(4):
            pos_in_group[np.flatnonzero(kGroupOf==g)] = np.arange(16)
        new_r = np.zeros((w,64), int); new2 = np.zeros((w,64), int); newrand = np.zeros((w,64), int)
        for l in range(64):
            t = 0; t2 = 0
            L = local[valid[:, l], l]; Cn = cnt[valid[:, l], l]
            nv = len(L)
            lrand = np.concatenate([rng.permutation(L[Cn==1]), rng.permutation(L[Cn==2]), rng.permutation(L[Cn>2])])
            newrand[:nv, l] = lrand & 15
            for ph in (1, 2, 3):
                sel = (Cn == ph) if ph < 3 else (Cn > 2)
                Lp = L[sel]
                buckets = [list(Lp[(Lp & 15) == q]) for q in range(16)]
                left = len(Lp)
                # scheme A: residue pointer advances by one each step (global step t), take next non-empty cyclically
                while left:
                    q = (t + pos_in_group[l]) & 15
                    k = 0
                    while not buckets[(q + k) & 15]: k += 1
                    b = buckets[(q + k) & 15]; b.pop(); new_r[t, l] = (q + k) & 15; t += 1; left -= 1
            # idle slots read row 0 -> residue 0
        c, r = cycles(new_r, allv); res_loc[0]+=c; res_loc[1]+=r
        c, r = cycles(newrand, allv); res_rand[0]+=c; res_rand[1]+=r
    print("side", side, "greedy %.3f  lane-local cyclic %.3f  random %.3f cycles per group-read" % (res_cur[0]/res_cur[1], res_loc[0]/res_loc[1], res_rand[0]/res_rand[1]))
