#!/usr/bin/env python3
"""profiles/ubench/r05/c5_partition.py (round 4's c5_one_partition.py + a host-stepped mode) -- ONE partition of BASELINE config C5 alone on the GPU (30 000 genes x the
first 25 000 of 200 000 cells, rank 20), as a local group of one: the device-driven partitioned loop exactly as an 8-GPU
rank runs it (k_update x2, gene-side sweep, k_pack, k_tail_h, the group sum where RCCL's all-reduce goes, cell-side sweep,
k_tail_data, k_control), with no other partition sharing the chip -- so rocprofv3's per-kernel times are the partition's
own (profiles/collect_r04_evidence.sh ran all eight partitions side by side: their kernels overlapped).
    rocprofv3 --kernel-trace --stats ... -- python3 profiles/ubench/r04/c5_one_partition.py [--steps 300]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--rccl", action="store_true", help="a ONE-rank RCCL communicator instead of the local group of one: the schedule "
                    "an RCCL rank runs (k_pack between the sweeps, collectives by librccl), with nothing to exchange")
    ap.add_argument("--host", action="store_true", help="with --rccl: host-stepped steps (step_local / allreduce / step_finish): the collective's "
                    "kernels run ALONE on the chip between the sweep and the closing kernel")
    args = ap.parse_args()
    import torch
    import bench
    import ccfindr_amd as C
    from ccfindr_amd import synth
    # the first partition's column block, generated once per box (the full 30 000 x 200 000 matrix takes a minute or two on
    # one core) and kept under /tmp for the script's later invocations
    import scipy.sparse as sp
    n, m, r, P = 30000, 200000, 20, 8
    cache = "/tmp/vbnmf_c5_block0.npz"
    if os.path.exists(cache):
        z = np.load(cache)
        Xb = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=(n, m // P))
    else:
        X, n, m, r = bench.make_c5(False)
        Xb = X.tocsc()[:, :m // P]
        np.savez(cache + ".tmp.npz", data=Xb.data, indices=Xb.indices, indptr=Xb.indptr)
        os.replace(cache + ".tmp.npz", cache)
    M = C.CountMatrix(Xb)
    cols = (0, m // P)
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    wh = synth.random_state(n, m, r, hy, seed=1005)
    comm = C.Communicator.rccl(C.Communicator.unique_id(), 1, 0, 0) if args.rccl else C.Communicator.local(1)
    eng = C.VBEngine(M, r, cols=cols, m_global=m)          # (M holds the block only: the engine's partition is all of it)
    eng.attach_comm(comm)
    eng.set_state(wh["lw"], wh["lh"][:, cols[0]:cols[1]], wh["eh"][:, cols[0]:cols[1]])
    if args.rccl:                              # an RCCL communicator is driven through its engine
        eng.allreduce(); eng.state_finish()
        run = eng.run
    else:
        comm.state_finish()
        run = comm.run
    if args.host:
        for _ in range(args.steps):
            eng.step_local(hy); eng.allreduce(); eng.step_finish()
        torch.cuda.synchronize()
        print(json.dumps({"mode": "host-stepped", "steps": args.steps}), flush=True)
        eng.close(); comm.close(); M.close()
        return
    run(hy, Itmax=50, Tol=0.0, flags=(False,) * 4)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(hy, Itmax=args.steps, Tol=0.0, flags=(False,) * 4)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / args.steps)
    info = eng.layout_info()
    nnz_local = int(Xb.nnz)
    out = {"workload": f"C5 partition 1 of {P}: {n} x {cols[1] - cols[0]} of {m} cells, nnz {nnz_local}, rank {r}, alone on the GPU",
           "communicator": "rccl, one rank" if args.rccl else "local group of one",
           "cell_order": os.environ.get("VBNMF_CELL_ORDER", "auto"), "ms_per_step": 1e3 * float(np.median(ts)),
           "tasks_gene": info["tasks_gene_side"], "tasks_cell": info["tasks_cell_side"]}
    print(json.dumps(out), flush=True)
    eng.close(); comm.close(); M.close()


if __name__ == "__main__":
    main()
