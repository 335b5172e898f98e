#!/usr/bin/env python3
"""tests/manual_c4_sweep.py (run by hand through gpurun) -- BASELINE.json config C4 on ONE GPU: vb_factorize over
ranks 2..20 on the C3 matrix (the units `vb_factorize_sharded` deals out longest-first over 8 GPUs), with the
reference's defaults (hyper-parameter updates on, Tol = 1e-5) and Itmax capped.  Reports wall time per rank (engine
build + initial state + device-driven loop + state download), iterations used and the log-evidence curve.
Writes gpurun_out/c4_sweep.json.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--itmax", type=int, default=2000)
    ap.add_argument("--ranks", default="2-20")
    args = ap.parse_args()
    import bench
    import ccfindr_amd as C
    lo, hi = (int(v) for v in args.ranks.split("-"))
    name, X, _ = bench.make_workload(False)
    t0 = time.perf_counter()
    M = C.CountMatrix(X)
    t_ingest = time.perf_counter() - t0
    rows = []
    t_all = time.perf_counter()
    for r in range(lo, hi + 1):
        t0 = time.perf_counter()
        res = C.vb_factorize(M, ranks=r, nrun=1, verbose=0, Itmax=args.itmax, Tol=1e-5, seed=7)
        dt = time.perf_counter() - t0
        rows.append({"rank": r, "seconds": dt, "iterations": res.nsteps[0], "lml": res.measure["lml"][0],
                     "aw": res.measure["aw"][0], "bw": res.measure["bw"][0], "ah": res.measure["ah"][0], "bh": res.measure["bh"][0]})
        print(rows[-1], flush=True)
    total = time.perf_counter() - t_all
    out = {"workload": name + f", ranks {lo}..{hi}, hyper updates on, Tol 1e-5, Itmax {args.itmax}", "ingest_seconds": t_ingest,
           "sweep_seconds": total, "ranks": rows, "best_rank_by_lml": max(rows, key=lambda q: q["lml"])["rank"]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c4_sweep.json"), "w"), indent=1)
    print(f"sweep of {hi - lo + 1} ranks: {total:.1f} s (+ {t_ingest:.1f} s ingestion); best rank by lml: {out['best_rank_by_lml']}")


if __name__ == "__main__":
    main()
