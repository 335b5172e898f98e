#!/usr/bin/env python3
"""tests/manual_c4_sharded.py (run by hand through gpurun) -- BASELINE.json config C4 the way the 8-GPU node runs it,
rehearsed with P = 1, 2, 4, 6 processes on the ONE test GPU (gloo; all processes share cuda:0, on the node each has its
own; the GPU box allows six processes on its card): vb_factorize_sharded over ranks 2..20 on the C3 matrix, reference
defaults, the (run, rank) units dealt longest-first (no data-path collective, reference R/bayesian.R:261-263, 316).

Process 0 alone ingests X (once, outside the timed call, as a user of vb_factorize would hold a CountMatrix) and cuts the
sweep's pair of layouts; the others pass mat=None, receive a shell + the layouts through /dev/shm and read the result
from the node's shared segment.  Per process: the call's wall time split into layouts (cut / wait / import), units (of
which device stepping) and the exchange of the per-unit records.  Writes gpurun_out/c4_sharded.json.

    python tests/manual_c4_sharded.py [--procs 1,2,4,6]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RANKS = list(range(2, 21))


def worker(rank, world, port, path, q, concurrent=1):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import scipy.sparse as sp
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ccfindr_amd as C
        from ccfindr_amd import parallel
        M, t_ingest = None, 0.0
        if rank == 0:
            z = np.load(path, mmap_mode="r")
            X = sp.csc_matrix((np.asarray(z["data"]), np.asarray(z["indices"]), np.asarray(z["indptr"])), shape=tuple(z["shape"]))
            t0 = time.perf_counter()
            M = C.CountMatrix(X)
            t_ingest = time.perf_counter() - t0
            M.prepare_async()                # ingested for whole-matrix factorisations: the second orientation forms in the background
        # time the device-driven loops of this process's units
        stepping = {"s": 0.0, "units": []}
        orig = C.VBEngine.run

        def timed_run(self, *a, **kw):
            t = time.perf_counter()
            out = orig(self, *a, **kw)
            dt = time.perf_counter() - t
            stepping["s"] += dt
            stepping["units"].append((self.rank, out["it"], dt))
            return out

        C.VBEngine.run = timed_run
        tm = {}
        dist.barrier()
        t0 = time.perf_counter()
        res = parallel.vb_factorize_sharded(M, ranks=RANKS, nrun=1, Itmax=2000, Tol=1e-5, seed=7, device=0, timings=tm, concurrent=concurrent)
        t_all = time.perf_counter() - t0
        dist.barrier()
        t_wall = time.perf_counter() - t0                     # until the slowest process has its result
        import hashlib
        digest = hashlib.sha256(b"".join(np.ascontiguousarray(b).tobytes() for b in res.basis + res.coeff + res.dbasis + res.dcoeff)).hexdigest()
        q.put({"process": rank, "ingest_s": t_ingest, "sharded_call_s": t_all, "wall_to_slowest_s": t_wall,
               "stepping_s": stepping["s"], "overhead_s": t_all - stepping["s"], "split": tm,
               "units": [u[0] for u in stepping["units"]], "iterations": int(sum(u[1] for u in stepping["units"])),
               "nsteps_all_ranks": list(res.nsteps), "lml": [float(v) for v in res.measure["lml"]], "result_sha256": digest})
    finally:
        dist.destroy_process_group()


def run_world(world, path, concurrent=1):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30900 + (os.getpid() % 500) + world
    procs = [ctx.Process(target=worker, args=(k, world, port, path, q, concurrent)) for k in range(world)]
    t0 = time.perf_counter()
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=900) for _ in procs], key=lambda o: o["process"])
    for p in procs:
        p.join(timeout=120)
    wall = time.perf_counter() - t0
    assert all(o["nsteps_all_ranks"] == outs[0]["nsteps_all_ranks"] and o["lml"] == outs[0]["lml"] and
               o["result_sha256"] == outs[0]["result_sha256"] for o in outs), "the processes disagree on the result"
    return {"processes": world, "concurrent_units_per_process": concurrent, "wall_s_including_process_start_and_import": wall,
            "sharded_call_s_max": max(o["sharded_call_s"] for o in outs),
            "overhead_s_max": max(o["overhead_s"] for o in outs), "per_process": outs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", default="1,2,4,6")
    ap.add_argument("--concurrent", type=int, default=4)
    args = ap.parse_args()
    import bench
    name, X, _ = bench.make_workload(False)
    path = os.path.join(ROOT, "gpurun_out", "c3_tmp.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez(path, data=X.data, indices=X.indices, indptr=X.indptr, shape=np.asarray(X.shape))
    rows = []
    for world in [int(v) for v in args.procs.split(",")]:
        row = run_world(world, path, args.concurrent)
        rows.append(row)
        print(f"P={world}: call {row['sharded_call_s_max']:.2f} s (slowest process), overhead beside stepping {row['overhead_s_max']:.2f} s", flush=True)
        for o in row["per_process"]:
            sp = o["split"]
            print(f"   process {o['process']}: call {o['sharded_call_s']:.3f}  stepping {o['stepping_s']:.3f}  layouts {sp['layout_s']:.3f}  "
                  f"units {sp['units_s']:.3f}  exchange {sp['gather_s']:.3f}  ({len(o['units'])} units: ranks {o['units']})", flush=True)
            print("      first call", round(sp["first_call_s"], 3), "layout detail", {k: round(v, 3) for k, v in (sp.get("layout_detail") or {}).items()}, flush=True)
            ut = sp.get("unit_detail") or []
            if ut:
                print("      units: " + "  ".join(f"r{u['rank']}: eng {u['engine_s']:.3f} draw {u['draw_s']:.3f} set {u['set_state_s']:.3f} "
                                                   f"loop {u['loop_s']:.3f} get {u['get_state_s']:.3f}" for u in ut), flush=True)
    os.remove(path)
    same = all(r["per_process"][0]["result_sha256"] == rows[0]["per_process"][0]["result_sha256"] for r in rows)
    assert same, "the result depends on the number of processes"
    out = {"workload": name + ", ranks 2..20 sharded over P processes on ONE GPU (gloo), reference defaults; process 0 holds X, "
                              "the others a shell; layouts and results through /dev/shm",
           "result_identical_for_every_P": same, "by_processes": rows}
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c4_sharded.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
