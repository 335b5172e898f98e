#!/usr/bin/env python3
"""tests/manual_c5_check.py (run by hand through gpurun; it lives under tests/ because it uses the CPU oracle as its checker) -- BASELINE.json config C5 on ONE GPU: 30 000 genes x 200 000 cells (~5 % stored), rank 20,
cells cut 8 ways.  The eight partition engines live side by side on the one device; their reduce buffers
[swsum | rowSums(eh) | scalars] are summed on the device where the 8-GPU run issues its RCCL all-reduce.

Checks   : lkh of the first steps against the stored-entries CPU restatement (oracle, OpenMP) on the whole matrix;
           the replicated gene-side state bit-identical across partitions.
Measures : per-partition step_local / step_finish time (each engine timed alone), i.e. the compute part of one
           8-GPU step; the all-reduce payload.  Writes gpurun_out/c5_check.json.

    python tests/manual_c5_check.py [--cells 200000] [--parts 8] [--steps 2] [--timing-steps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # tests/ -> repo root
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=30000)
    ap.add_argument("--cells", type=int, default=200000)
    ap.add_argument("--rank", type=int, default=20)
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--timing-steps", type=int, default=20)
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()

    import torch
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from ccfindr_amd.parallel import cell_partition

    n, m, r, P = args.genes, args.cells, args.rank, args.parts
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    t0 = time.time()
    k = 20
    # 5 % stored at 30 000 genes needs deeper cells than C3's 1500 counts: log-normal around 1950, alpha0 = 0.1
    depth = np.round(np.random.default_rng(5).lognormal(np.log(1950.0), 0.3, size=m)).astype(np.int64)
    X = synth.fill_empty(synth.simulate_data(n, [m // k] * k, alpha0=0.1, seed=5, depth=depth), seed=5)
    print(f"matrix {n} x {m}, nnz {X.nnz} ({100.0 * X.nnz / n / m:.2f} %), generated in {time.time() - t0:.1f} s", flush=True)
    t0 = time.time()
    M = C.CountMatrix(X)
    print(f"ingested in {time.time() - t0:.1f} s", flush=True)
    wh = synth.random_state(n, m, r, hy, seed=1005)
    cuts = cell_partition(m, P)
    t0 = time.time()
    parts = [C.VBEngine(M, r, cols=c, m_global=m) for c in cuts]
    print(f"{P} partition engines (layouts) built in {time.time() - t0:.1f} s; device memory {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB free", flush=True)
    reds = [p.reduce_tensor() for p in parts]

    def allreduce():
        torch.cuda.synchronize()
        s = reds[0].clone()
        for q in reds[1:]:
            s += q
        for q in reds:
            q.copy_(s)
        torch.cuda.synchronize()

    for p, (b, e) in zip(parts, cuts):
        p.set_state(wh["lw"], wh["lh"][:, b:e], wh["eh"][:, b:e])
    allreduce()
    for p in parts:
        p.state_finish()

    lk = []
    for _ in range(args.steps):
        for p in parts:
            p.step_local(hy)
        allreduce()
        outs = [p.step_finish() for p in parts]
        assert all(o == outs[0] for o in outs), "replicated scalars differ between partitions"
        lk.append(outs[0][0])
        print(f"step {len(lk)}: lkh = {lk[-1]:.15g}", flush=True)
    a, b = parts[0].get_state(("lw", "ew")), parts[-1].get_state(("lw", "ew"))
    assert np.array_equal(a["lw"], b["lw"]) and np.array_equal(a["ew"], b["ew"]), "gene-side state is not replicated bit for bit"

    out = {"workload": f"C5 {n} x {m}, rank {r}, {P} cell partitions on one GPU", "nnz": int(X.nnz), "lkh": lk,
           "reduce_doubles": int(reds[0].numel()), "reduce_bytes": int(reds[0].numel() * 8)}
    if not args.no_oracle:
        from oracle import vbnmf_oracle as O
        S = X.tocsc()
        nt = min(16, len(os.sched_getaffinity(0)))
        ref, want = wh, []
        t0 = time.time()
        for _ in range(args.steps):
            ref = O.update_csc(n, m, S.indptr, S.indices, S.data, ref, hy, nthreads=nt)
            want.append(ref["lkh"])
        out["oracle_lkh"] = want
        out["oracle_s_per_step"] = (time.time() - t0) / args.steps
        out["oracle_threads"] = nt
        err = max(abs(g / w - 1) for g, w in zip(lk, want))
        out["lkh_rel_err"] = err
        print(f"oracle ({nt} threads, {out['oracle_s_per_step']:.1f} s/step): max lkh rel err {err:.3e}", flush=True)
        assert err <= 1e-10, err
        ew_err = float(np.max(np.abs(a["ew"] - ref["ew"]) / np.abs(ref["ew"])))
        out["ew_rel_err"] = ew_err
        assert ew_err <= 1e-10, ew_err

    # the compute part of one 8-GPU step: every partition timed alone (step_local + step_finish, reduce buffer left as is)
    times = []
    for p in parts:
        for _ in range(3):
            p.step_local(hy); p.step_finish()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.timing_steps):
            p.step_local(hy); p.step_finish()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / args.timing_steps)
    out["partition_ms_per_step"] = [1e3 * t for t in times]
    out["slowest_partition_ms"] = 1e3 * max(times)
    out["projected_8gpu_it_per_s_compute_only"] = 1.0 / max(times)
    print(f"per-partition step (local + finish): {[round(1e3 * t, 3) for t in times]} ms", flush=True)
    for p in parts:
        p.close()

    # The same partitions under the DEVICE-DRIVEN group loop (vbnmf_group_run: gene-side sweep, k_pack, the n x r sum on a
    # second stream beside the cell-side sweep, the small sum of the evidence partials, the folded control step -- all queued from C++): the P partitions
    # share this one GPU, so a group step is P partition steps back to back; per partition = group step / P.
    comm = C.Communicator.local(P)
    parts = [C.VBEngine(M, r, cols=c, m_global=m) for c in cuts]
    for p, (b, e) in zip(parts, cuts):
        p.attach_comm(comm)
        p.set_state(wh["lw"], wh["lh"][:, b:e], wh["eh"][:, b:e])
    comm.state_finish()
    first = comm.run(hy, Itmax=args.steps, Tol=0.0, n0=10, dn=1, flags=(False,) * 4, history=True)
    out["group_loop_lkh"] = [float(v) for v in first["history"][:, 0]]
    out["group_loop_lkh_rel_err_vs_host_stepped"] = max(abs(g / w - 1) for g, w in zip(out["group_loop_lkh"], lk))
    assert out["group_loop_lkh_rel_err_vs_host_stepped"] <= 1e-11, out["group_loop_lkh_rel_err_vs_host_stepped"]
    K = max(args.timing_steps, 20)
    comm.run(hy, Itmax=5, Tol=0.0, n0=10, dn=1, flags=(False,) * 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    comm.run(hy, Itmax=K, Tol=0.0, n0=10, dn=1, flags=(False,) * 4)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    out["group_loop_ms_per_group_step"] = 1e3 * dt
    out["group_loop_ms_per_partition_step"] = 1e3 * dt / P
    print(f"device-driven group loop: {1e3 * dt:.3f} ms per group step = {1e3 * dt / P:.3f} ms per partition step "
          f"(host-stepped: {1e3 * max(times):.3f})", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c5_check.json"), "w"), indent=1)
    for p in parts:
        p.close()
    comm.close()
    print("c5 check ok", flush=True)


if __name__ == "__main__":
    main()
