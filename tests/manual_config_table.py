#!/usr/bin/env python3
"""tests/manual_config_table.py (run by hand through gpurun; it lives under tests/ because it uses the CPU oracle as its checker) -- the BASELINE.json configurations C1..C3 side by side:
GPU iterations/s of the device-driven loop, the literal dense CPU restatement on ONE thread (faithful to the
reference's src/Makevars: no OpenMP) and the stored-entries CPU restatement on the box's cores.  The dense literal
form is timed at C1, C2 and, for C3, at the stated down-scale 20 000 x 5 000 (the first 5 000 cells) and reported per
matrix element.  Writes gpurun_out/configs.json.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # tests/ -> repo root
sys.path.insert(0, ROOT)
HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}


def gpu_rate(X, r, steps):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    n, m = X.shape
    eng = C.VBEngine(C.CountMatrix(X), r)
    wh = synth.random_state(n, m, r, HY, seed=1000 + r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    for _ in range(10):
        lkh, _ = eng.step(HY)
    t0 = time.perf_counter()
    res = eng.run(HY, Itmax=steps, Tol=0.0, n0=10, dn=1, flags=(False,) * 4)
    dt = time.perf_counter() - t0
    eng.close()
    return res["it"] / dt, wh


def main():
    import bench
    from ccfindr_amd import synth
    from oracle import vbnmf_oracle as O
    cores = bench.usable_cores()
    out = {"cores": cores, "configs": {}}

    def literal(X, wh, reps):
        A = np.asfortranarray(X.toarray() if hasattr(X, "toarray") else X)
        t0 = time.perf_counter()
        for _ in range(reps):
            wh = O.update_dense(A, wh, HY)
        return reps / (time.perf_counter() - t0), wh["lkh"]

    def sparse(X, wh, reps):
        S = X.tocsc() if hasattr(X, "tocsc") else __import__("scipy.sparse").sparse.csc_matrix(X)
        n, m = S.shape
        t0 = time.perf_counter()
        for _ in range(reps):
            wh = O.update_csc(n, m, S.indptr, S.indices, S.data, wh, HY, nthreads=cores)
        return reps / (time.perf_counter() - t0), wh["lkh"]

    # C1: simulate_data defaults, 200 x 500, rank 3
    X = synth.drop_empty(synth.simulate_data(200, (100, 150, 250), seed=1, sparse=False))
    g, wh = gpu_rate(X, 3, 2000)
    l, _ = literal(X, wh, 20)
    s, _ = sparse(X, wh, 50)
    out["configs"]["C1 200 x 500 rank 3"] = {"gpu_it_s": g, "cpu_literal_dense_1thread_it_s": l, "cpu_sparse_openmp_it_s": s}
    print("C1", out["configs"]["C1 200 x 500 rank 3"], flush=True)

    # C2: 2 000 x 10 000 dense counts (mean 2), rank 5
    X = synth.fill_empty(synth.simulate_data(2000, [2000] * 5, alpha0=2.0, seed=2, depth=np.full(10000, 4000)), seed=2)
    g, wh = gpu_rate(X, 5, 2000)
    l, _ = literal(X, wh, 2)
    s, _ = sparse(X, wh, 5)
    out["configs"]["C2 2000 x 10000 dense rank 5"] = {"nnz_fraction": X.nnz / 2e7, "gpu_it_s": g, "cpu_literal_dense_1thread_it_s": l,
                                                      "cpu_sparse_openmp_it_s": s}
    print("C2", out["configs"]["C2 2000 x 10000 dense rank 5"], flush=True)

    # C3: the headline; literal dense form at the down-scale 20 000 x 5 000 (first 5 000 cells), per element
    name, X, r = bench.make_workload(False)
    n, m = X.shape
    g, wh = gpu_rate(X, r, 1000)
    s, _ = sparse(X, wh, 3)
    Xd = X.tocsc()[:, :5000]
    whd = {k: (v[:, :5000] if k in ("lh", "eh") else v) for k, v in wh.items() if k in ("lw", "lh", "eh")}
    ld, _ = literal(Xd, whd, 1)
    out["configs"]["C3 20000 x 50000 sparse rank 10"] = {
        "gpu_it_s": g, "cpu_sparse_openmp_it_s": s, "cpu_literal_dense_1thread_it_s_at_20000x5000": ld,
        "cpu_literal_dense_1thread_ns_per_element": 1e9 / ld / (n * 5000),
        "cpu_literal_dense_1thread_it_s_extrapolated_to_C3": ld * 5000 / m}
    print("C3", out["configs"]["C3 20000 x 50000 sparse rank 10"], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
