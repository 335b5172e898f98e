#!/usr/bin/env python3
"""tests/manual_c4_sweep.py (run by hand through gpurun) -- BASELINE.json config C4 on ONE GPU: vb_factorize over
ranks 2..20 on the C3 matrix (the units `vb_factorize_sharded` deals out longest-first over 8 GPUs), with the
reference's defaults (hyper-parameter updates on, Tol = 1e-5) and Itmax capped.

Two measurements, both written to gpurun_out/c4_sweep.json:
  * `sweep_seconds`: ONE call vb_factorize(M, ranks=2..20) end to end (what a user of the reference's API runs);
  * `ranks`: the same units taken apart -- per rank `setup_s` (engine creation incl. any layout cut + initial state +
    priming sweep + state download) and `stepping_s` (the device-driven loop alone), iterations and log evidence.
`--classes 0` switches the rank classes off (every LDS row size cuts its own pair of layouts: round 2's behaviour).
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
    ap.add_argument("--classes", type=int, default=1)
    args = ap.parse_args()
    import bench
    import ccfindr_amd as C
    from ccfindr_amd import bayesian
    lo, hi = (int(v) for v in args.ranks.split("-"))
    ranks = list(range(lo, hi + 1))
    name, X, _ = bench.make_workload(False)
    n, m = X.shape
    t0 = time.perf_counter()
    M = C.CountMatrix(X)
    t_ingest = time.perf_counter() - t0

    # (1) the call a user makes
    t0 = time.perf_counter()
    res = C.vb_factorize(M, ranks=ranks, nrun=1, verbose=0, Itmax=args.itmax, Tol=1e-5, seed=7, geometry_classes=args.classes)
    sweep_s = time.perf_counter() - t0
    print(f"vb_factorize(ranks={lo}..{hi}): {sweep_s:.2f} s, iterations {sum(res.nsteps)}", flush=True)
    M.close()

    # (2) the same units taken apart, on a fresh matrix handle (nothing cached)
    M = C.CountMatrix(X)
    if args.classes:
        M.plan_ranks(ranks, args.classes)
    hyper0 = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    rows = []
    t_all = time.perf_counter()
    for r in ranks:
        rng = np.random.default_rng([7, 1, r])
        t0 = time.perf_counter()
        eng = C.VBEngine(M, r)
        t1 = time.perf_counter()
        wh0 = bayesian.vb_init(n, m, M, r, hyper=dict(hyper0), initializer="random", rng=rng)
        eng.set_state(wh0["lw"], wh0["lh"], wh0["eh"])
        t2 = time.perf_counter()
        out = eng.run(dict(hyper0), Itmax=args.itmax, Tol=1e-5, n0=10, dn=1, flags=(True,) * 4)
        t3 = time.perf_counter()
        eng.get_state(("ew", "eh", "dw", "dh"))
        eng.close()
        t4 = time.perf_counter()
        rows.append({"rank": r, "seconds": t4 - t0, "setup_s": (t2 - t0) + (t4 - t3), "engine_create_s": t1 - t0,
                     "init_state_s": t2 - t1, "stepping_s": t3 - t2, "download_close_s": t4 - t3,
                     "iterations": out["it"], "lml": out["lk0"]})
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in rows[-1].items()}, flush=True)
    total = time.perf_counter() - t_all
    M.close()
    assert [q["iterations"] for q in rows] == list(res.nsteps), "the two passes ran different iteration counts"
    out = {"workload": name + f", ranks {lo}..{hi}, hyper updates on, Tol 1e-5, Itmax {args.itmax}",
           "geometry_classes": args.classes, "ingest_seconds": t_ingest,
           "sweep_seconds": sweep_s, "sweep_iterations": int(sum(res.nsteps)),
           "taken_apart_seconds": total, "setup_seconds": sum(q["setup_s"] for q in rows),
           "stepping_seconds": sum(q["stepping_s"] for q in rows), "ranks": rows,
           "best_rank_by_lml": max(rows, key=lambda q: q["lml"])["rank"]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c4_sweep.json"), "w"), indent=1)
    print(f"sweep of {hi - lo + 1} ranks: {sweep_s:.2f} s in one call; taken apart {total:.2f} s = setup {out['setup_seconds']:.2f} "
          f"+ stepping {out['stepping_seconds']:.2f} (+ {t_ingest:.1f} s ingestion); best rank by lml: {out['best_rank_by_lml']}")


if __name__ == "__main__":
    main()
