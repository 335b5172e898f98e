import numpy as np, scipy.sparse as sp
X = sp.load_npz('/tmp/c3.npz').tocsc()
n,m = X.shape
R = X.tocsr()
rl = np.diff(R.indptr); cl = np.diff(X.indptr)
print("gene row len: mean %.0f median %.0f p10 %.0f p90 %.0f max %d" % (rl.mean(), np.median(rl), *np.percentile(rl,[10,90]), rl.max()))
print("cell col len: mean %.0f median %.0f p10 %.0f p90 %.0f max %d" % (cl.mean(), np.median(cl), *np.percentile(cl,[10,90]), cl.max()))
# entries share by gene-length deciles
srt = np.sort(rl)[::-1]; cs = np.cumsum(srt)/srt.sum()
for f in (0.01,0.05,0.1,0.2,0.5): print("top %.0f%% genes hold %.2f of entries" % (100*f, cs[int(f*n)-1]))
def tasks(Mcsr, nblk_bounds, cap=256):
    # count tasks = sum over (major, block) ceil(cnt/cap); returns n_tasks, n_pairs, len histogram
    nmaj = Mcsr.shape[0]
    blk = np.searchsorted(nblk_bounds, Mcsr.indices, side='right')-1
    nb = len(nblk_bounds)-1
    maj = np.repeat(np.arange(nmaj), np.diff(Mcsr.indptr))
    key = maj.astype(np.int64)*nb + blk
    cnt = np.bincount(key, minlength=nmaj*nb)
    cnt = cnt[cnt>0]
    pieces = (cnt+cap-1)//cap
    return pieces.sum(), len(cnt), cnt
# current: equal-entry blocks approximated by equal width
for side,(M,nb) in enumerate(((R,26),(X.T.tocsr(),11))):
    nmin = M.shape[1]
    b = np.linspace(0,nmin,nb+1).astype(int)
    nt, npairs, cnt = tasks(M,b)
    print("side",side,"blocks",nb,"tasks",nt,"pairs",npairs,"mean len %.1f"%(M.nnz/nt), "pairs<=4: %.3f <=16: %.3f <=64 %.3f"%((cnt<=4).mean(),(cnt<=16).mean(),(cnt<=64).mean()),
          "entries in pairs<=16: %.3f <=64: %.3f"%(cnt[cnt<=16].sum()/M.nnz, cnt[cnt<=64].sum()/M.nnz))
