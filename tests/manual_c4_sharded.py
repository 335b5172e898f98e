#!/usr/bin/env python3
"""tests/manual_c4_sharded.py (run by hand through gpurun) -- BASELINE.json config C4 the way the 8-GPU node runs it,
rehearsed with TWO processes on the one test GPU (gloo for the result gather; both processes share cuda:0, on the node
each has its own): vb_factorize_sharded over ranks 2..20 on the C3 matrix, reference defaults, the (run, rank) units
dealt longest-first (no data-path collective, reference R/bayesian.R:261-263, 316).  Per process: wall time split into
matrix ingestion, its units (of which device stepping), and the gather.  Writes gpurun_out/c4_sharded.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, path, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import scipy.sparse as sp
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ccfindr_amd as C
        from ccfindr_amd import bayesian, parallel
        z = np.load(path, mmap_mode="r")
        X = sp.csc_matrix((np.asarray(z["data"]), np.asarray(z["indices"]), np.asarray(z["indptr"])), shape=tuple(z["shape"]))
        t0 = time.perf_counter()
        M = C.CountMatrix(X)
        t_ingest = time.perf_counter() - t0
        # time the device-driven loops of this process's units
        stepping = {"s": 0.0, "units": []}
        orig = C.VBEngine.run

        def timed_run(self, *a, **kw):
            t = time.perf_counter()
            out = orig(self, *a, **kw)
            dt = time.perf_counter() - t
            stepping["s"] += dt
            stepping["units"].append((self.rank if hasattr(self, "rank") else None, out["it"], dt))
            return out

        C.VBEngine.run = timed_run
        dist.barrier()
        t0 = time.perf_counter()
        res = parallel.vb_factorize_sharded(M, ranks=list(range(2, 21)), nrun=1, Itmax=2000, Tol=1e-5, seed=7, device=0)
        t_all = time.perf_counter() - t0
        q.put({"process": rank, "ingest_s": t_ingest, "sharded_call_s": t_all, "stepping_s": stepping["s"],
               "units": len(stepping["units"]), "iterations": int(sum(u[1] for u in stepping["units"])),
               "nsteps_all_ranks": list(res.nsteps), "lml": [float(v) for v in res.measure["lml"]]})
    finally:
        dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    import bench
    name, X, _ = bench.make_workload(False)
    path = os.path.join(ROOT, "gpurun_out", "c3_tmp.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez(path, data=X.data, indices=X.indices, indptr=X.indptr, shape=np.asarray(X.shape))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30900 + (os.getpid() % 500)
    world = 2
    procs = [ctx.Process(target=worker, args=(k, world, port, path, q)) for k in range(world)]
    t0 = time.perf_counter()
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=900) for _ in procs], key=lambda o: o["process"])
    for p in procs:
        p.join(timeout=120)
    wall = time.perf_counter() - t0
    os.remove(path)
    assert outs[0]["nsteps_all_ranks"] == outs[1]["nsteps_all_ranks"] and outs[0]["lml"] == outs[1]["lml"]
    out = {"workload": name + ", ranks 2..20 sharded over 2 processes on ONE GPU (gloo gather), reference defaults",
           "wall_s_including_process_start_and_import": wall, "processes": outs}
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c4_sharded.json"), "w"), indent=1)
    for o in outs:
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in o.items() if k not in ("nsteps_all_ranks", "lml")})
    print(f"wall {wall:.1f} s")


if __name__ == "__main__":
    main()
