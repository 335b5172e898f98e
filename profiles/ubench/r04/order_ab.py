#!/usr/bin/env python3
"""profiles/ubench/r04/order_ab.py -- VERDICT r03 Next #5, measured before anything is built into the layout: the step is
permutation-equivariant, so a LABEL-FREE renumbering of the cells (and genes) can be tried by permuting X before
ingestion.  Cells: gene-hash sketch (d groups) -> spherical k-means (K centroids, a few iterations) -> clusters chained by
centroid similarity -> cells sorted by cluster.  Genes (optional): by the cell cluster that holds most of their entries.
Per variant and rank: (major, block) tasks per side, k_sweep / step time, it/s.  One box, variants interleaved.
    python3 profiles/ubench/r04/order_ab.py [--ranks 10,20] [--steps 400]"""
import argparse, json, os, sys, time
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)


def cell_order(X, d=64, K=32, iters=6, seed=1):
    n, m = X.shape
    rng = np.random.default_rng(seed)
    grp = rng.integers(0, d, size=n)
    G = sp.csr_matrix((np.ones(n), (np.arange(n), grp)), shape=(n, d))
    B = X.copy(); B.data = np.sqrt(B.data)
    S = np.asarray((B.T @ G).todense())                      # m x d
    S /= np.maximum(np.linalg.norm(S, axis=1, keepdims=True), 1e-300)
    C = S[rng.choice(m, K, replace=False)].copy()
    for _ in range(iters):
        lab = np.argmax(S @ C.T, axis=1)
        for k in range(K):
            sel = lab == k
            if sel.any():
                c = S[sel].sum(0); C[k] = c / max(np.linalg.norm(c), 1e-300)
    lab = np.argmax(S @ C.T, axis=1)
    # chain the clusters: nearest unvisited centroid next
    sim = C @ C.T
    order = [int(np.argmax(np.bincount(lab, minlength=K)))]
    left = set(range(K)) - set(order)
    while left:
        cur = order[-1]
        nxt = max(left, key=lambda k: sim[cur, k])
        order.append(nxt); left.remove(nxt)
    pos = np.empty(K, int); pos[order] = np.arange(K)
    return np.argsort(pos[lab], kind="stable"), lab


def gene_order(X, cell_lab, K):
    n, m = X.shape
    H = sp.csr_matrix((np.ones(m), (np.arange(m), cell_lab)), shape=(m, K))
    B = X.copy(); B.data[:] = 1.0
    Gk = np.asarray((B @ H).todense())                       # n x K entries of each gene per cell cluster
    share = Gk / np.maximum(Gk.sum(0, keepdims=True), 1)
    dom = share.argmax(1)
    return np.lexsort((-Gk.max(1), dom))


def run(X, r, steps, tag):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    import torch
    HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    n, m = X.shape
    t0 = time.perf_counter(); M = C.CountMatrix(X); t_ing = time.perf_counter() - t0
    t0 = time.perf_counter(); eng = C.VBEngine(M, r); t_eng = time.perf_counter() - t0
    wh = synth.random_state(n, m, r, HY, seed=1003)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    eng.run(HY, Itmax=600, Tol=0.0, flags=(False,) * 4)
    eng.timing_enable(True)
    for _ in range(100):
        lk, _ = eng.step(HY)
    ms, cnt = eng.timing_get()
    eng.timing_enable(False)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = eng.run(HY, Itmax=steps, Tol=0.0, flags=(False,) * 4)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    info = eng.layout_info()
    out = {"variant": tag, "rank": r, "it_per_s": steps / float(np.median(ts)), "step_us": 1e6 * float(np.median(ts)) / steps,
           "k_sweep_us": 1e3 * ms / cnt, "tasks_gene": info["tasks_gene_side"], "tasks_cell": info["tasks_cell_side"],
           "slots_gene": info["slots_gene_side"], "slots_cell": info["slots_cell_side"], "engine_create_s": t_eng, "lkh": res["lkh"]}
    print(json.dumps(out), flush=True)
    eng.close(); M.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="10,20")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--K", type=int, default=32)
    ap.add_argument("--d", type=int, default=64)
    args = ap.parse_args()
    import bench
    name, X, _ = bench.make_workload(False)
    X = X.tocsc()
    t0 = time.perf_counter(); co, lab = cell_order(X, d=args.d, K=args.K); t_co = time.perf_counter() - t0
    print(f"cell order (numpy, d={args.d}, K={args.K}): {t_co:.2f} s; cluster sizes {np.bincount(lab, minlength=args.K).tolist()}", flush=True)
    Xc = X[:, co].tocsc()
    go = gene_order(Xc, lab[co], args.K)
    Xcg = Xc[go, :].tocsc()
    rows = []
    for r in [int(v) for v in args.ranks.split(",")]:
        for rep in range(2):
            for tag, A in (("orig", X), ("cells", Xc), ("cells+genes", Xcg)):
                rows.append(run(A, r, args.steps, tag))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r04_order_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
