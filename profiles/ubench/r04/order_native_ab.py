#!/usr/bin/env python3
"""profiles/ubench/r04/order_native_ab.py -- the layout's own cell ordering (csrc/order.cpp) off / on, same box,
interleaved: VBNMF_CELL_ORDER=0 / 1 is read when a CountMatrix first needs its order, so every variant ingests afresh.
Per variant and rank: tasks per side, engine creation (includes the ordering + both layouts), k_sweep, step, it/s."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}


def run(X, r, steps, mode):
    import ccfindr_amd as C
    from ccfindr_amd import synth
    import torch
    os.environ["VBNMF_CELL_ORDER"] = mode
    n, m = X.shape
    M = C.CountMatrix(X)
    t0 = time.perf_counter(); eng = C.VBEngine(M, r); t_eng = time.perf_counter() - t0
    wh = synth.random_state(n, m, r, HY, seed=1003)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    eng.run(HY, Itmax=600, Tol=0.0, flags=(False,) * 4)
    eng.timing_enable(True)
    for _ in range(100):
        eng.step(HY)
    ms, cnt = eng.timing_get()
    eng.timing_enable(False)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = eng.run(HY, Itmax=steps, Tol=0.0, flags=(False,) * 4)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    info = eng.layout_info()
    out = {"order": mode, "rank": r, "it_per_s": steps / float(np.median(ts)), "step_us": 1e6 * float(np.median(ts)) / steps,
           "k_sweep_us": 1e3 * ms / cnt, "tasks_gene": info["tasks_gene_side"], "tasks_cell": info["tasks_cell_side"],
           "engine_create_s": t_eng, "lkh": res["lkh"]}
    print(json.dumps(out), flush=True)
    eng.close(); M.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="10,20")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--modes", default="0,1")
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import bench
    name, X, _ = bench.make_workload(False)
    rows = []
    for r in [int(v) for v in args.ranks.split(",")]:
        for rep in range(args.reps):
            for mode in args.modes.split(","):
                rows.append(run(X, r, args.steps, mode))
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r04_order_native_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
