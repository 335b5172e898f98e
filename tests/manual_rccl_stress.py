#!/usr/bin/env python3
"""tests/manual_rccl_stress.py (run by hand through gpurun) -- the library's multi-rank RCCL protocol under load: P processes
on the one GPU through the librccl stand-in (tests/fake_rccl, VBNMF_RCCL_LIB), a mid-size matrix, REPS device-driven runs of
STEPS steps each with hyper-parameter updates on (every step queues two collectives -- the n x r one on the comm stream beside
the cell-side sweep, the evidence slots on the main stream behind it -- and cycles the event ring), then a convergence run.  Every run's history
must be identical on all ranks and equal to the single engine's to 1e-10; the gene-side state bit-identical across ranks.
Writes gpurun_out/rccl_stress.json.      python tests/manual_rccl_stress.py [--procs 2] [--reps 6] [--steps 400]"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
FAKE = os.path.join(HERE, "fake_rccl", "_build", "libfake_rccl.so")
HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}


def problem():
    from ccfindr_amd import synth
    X = synth.fill_empty(synth.simulate_data(4000, [3000, 2500, 3500, 3000], alpha0=0.1, seed=41, depth=np.full(12000, 300)), seed=41)
    n, m = X.shape
    return X, n, m, 12, synth.random_state(n, m, 12, HY, seed=7)


def worker(rank, world, port, reps, steps, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VBNMF_RCCL_LIB"] = FAKE
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ccfindr_amd as C
        from ccfindr_amd.parallel import CellPartitionedEngine
        X, n, m, r, wh = problem()
        M = C.CountMatrix(X)
        eng = CellPartitionedEngine(M, r, device=0, native=True)
        outs = []
        t0 = time.perf_counter()
        for rep in range(reps):
            eng.set_state(wh["lw"], wh["lh"], wh["eh"])
            out = eng.run(HY, Itmax=steps, Tol=0.0, n0=10, dn=1, flags=(True,) * 4, history=True)
            outs.append((out["it"], out["reason"], out["history"].copy()))
        dt = time.perf_counter() - t0
        eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        conv = eng.run(HY, Itmax=5000, Tol=1e-6, n0=10, dn=1, flags=(True,) * 4, history=True)
        lw = eng.engine.get_state(("lw",))["lw"]
        q.put((rank, outs, (conv["it"], conv["reason"], conv["history"].copy()), lw, dt))
        eng.close()
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=2)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--steps", type=int, default=400)
    args = ap.parse_args()
    import torch.multiprocessing as mp
    import ccfindr_amd as C
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 34100 + (os.getpid() % 1000)
    procs = [ctx.Process(target=worker, args=(k, args.procs, port, args.reps, args.steps, q)) for k in range(args.procs)]
    for p in procs:
        p.start()
    X, n, m, r, wh = problem()
    whole = C.VBEngine(C.CountMatrix(X), r)
    whole.set_state(wh["lw"], wh["lh"], wh["eh"])
    want = whole.run(HY, Itmax=args.steps, Tol=0.0, n0=10, dn=1, flags=(True,) * 4, history=True)
    whole.set_state(wh["lw"], wh["lh"], wh["eh"])
    wconv = whole.run(HY, Itmax=5000, Tol=1e-6, n0=10, dn=1, flags=(True,) * 4, history=True)
    outs = sorted([q.get(timeout=900) for _ in procs], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
    worst = 0.0
    for rank, runs, conv, lw, dt in outs:
        for it, reason, hist in runs:
            assert it == args.steps and reason == 4
            assert np.array_equal(hist, outs[0][1][0][2]), "a run's history differs between ranks or between repetitions"
            worst = max(worst, rel(hist, want["history"]))
        assert conv[0] == outs[0][2][0] == wconv["it"] and conv[1] == wconv["reason"], (conv[0], wconv["it"])
        assert np.array_equal(conv[2], outs[0][2][2])
        worst = max(worst, rel(conv[2], wconv["history"]))
        assert np.array_equal(lw, outs[0][3])
    assert worst <= 1e-10, worst
    out = {"workload": f"{n} x {m}, rank {r}, cells partitioned over {args.procs} processes on one GPU, librccl stand-in",
           "runs": args.reps, "steps_per_run": args.steps, "collectives_per_rank": 2 * args.reps * args.steps,
           "convergence_run_steps": int(wconv["it"]), "worst_history_rel_err_vs_single_engine": worst,
           "seconds_for_the_runs_by_rank": [o[4] for o in outs]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rccl_stress.json"), "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
