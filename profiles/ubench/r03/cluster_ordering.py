import numpy as np, scipy.sparse as sp
X = sp.load_npz('/tmp/c3.npz').tocsc()
n,m = X.shape
rng = np.random.default_rng(3)
perm = rng.permutation(m)
lab = np.empty(m, int)
for k in range(10): lab[perm[k*5000:(k+1)*5000]] = k
def tasks(Mcsr, bounds, cap=256):
    nmaj = Mcsr.shape[0]
    blk = np.searchsorted(bounds, Mcsr.indices, side='right')-1
    nb = len(bounds)-1
    maj = np.repeat(np.arange(nmaj), np.diff(Mcsr.indptr))
    cnt = np.bincount(maj.astype(np.int64)*nb + blk, minlength=nmaj*nb)
    cnt = cnt[cnt>0]
    return ((cnt+cap-1)//cap).sum(), len(cnt), cnt
def report(tag, M, nb, cap=256):
    b = np.linspace(0,M.shape[1],nb+1).astype(int)
    nt,npairs,cnt = tasks(M,b,cap)
    print(tag, "cap",cap,"tasks",nt,"pairs",npairs,"mean len %.1f"%(M.nnz/nt), "entries in pairs<=16: %.3f <=64: %.3f"%(cnt[cnt<=16].sum()/M.nnz, cnt[cnt<=64].sum()/M.nnz))
# ideal: cells sorted by true cluster
co = np.argsort(lab, kind='stable')
Xc = X[:, co]
# genes by dominant cluster
G = np.zeros((n,10))
Bin = X.copy(); Bin.data[:] = 1
for k in range(10): G[:,k] = np.asarray(Bin[:, lab==k].sum(axis=1)).ravel()
dom = G.argmax(1)
go = np.lexsort((-G.max(1), dom))
Xcg = Xc[go,:]
for cap in (256,512):
    report("gene side, orig     ", X.tocsr(), 26, cap)
    report("gene side, cells by cluster", Xc.tocsr(), 26, cap)
    report("cell side, orig     ", X.T.tocsr(), 11, cap)
    report("cell side, genes by dom cluster", Xcg.T.tocsr(), 11, cap)
# how concentrated: fraction of a cell's entries in its top-2 gene blocks after gene ordering
M = Xcg.T.tocsr(); nb=11; b=np.linspace(0,n,nb+1).astype(int)
blk = np.searchsorted(b, M.indices, side='right')-1
maj = np.repeat(np.arange(m), np.diff(M.indptr))
cnt = np.bincount(maj.astype(np.int64)*nb+blk, minlength=m*nb).reshape(m,nb)
s = np.sort(cnt,1)[:,::-1]
print("cell side: mean share of entries in top1/top2/top3 blocks:", (s[:,0]/s.sum(1)).mean(), (s[:,:2].sum(1)/s.sum(1)).mean(), (s[:,:3].sum(1)/s.sum(1)).mean())
print("G share of dominant cluster per gene (entry-weighted):", (G.max(1).sum()/G.sum()))
