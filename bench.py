#!/usr/bin/env python3
"""bench.py -- VB-NMF update iterations/sec on the BASELINE.json headline workload.

Workload (config C3 of BASELINE.json / SURVEY.md section 8d): a 20 000-gene x 50 000-cell synthetic
sparse count matrix (Dirichlet-multinomial cluster mixture, ~5 % non-zero), rank 10,
hyper-parameters fixed at aw=bw=ah=bh=1, fudge = double eps.  A "step" is one full
vbnmf_update iteration (reference src/vbnmf_update.cpp:33-90, evidence included) on the
device-resident state, with lkh and the four hyper statistics delivered to the host for every
iteration, as the reference's caller needs them (reference R/bayesian.R:345-347).

The K timed steps are measured on two loops, each as the MEDIAN of 5 back-to-back repeats of the K-step region (every
repeat bracketed by a barrier + device synchronise; all five times are reported):
  * driven by the device (vbnmf_engine_run: the loop of vb_iterate with hyper_update and the stopping rule on the GPU,
    steps queued ahead, the per-step history -- lkh + 4 statistics of EVERY step -- written by the control kernel
    straight into pinned host memory) -> `value`; this is the path ccfindr_amd.vb_factorize runs by default;
  * stepped from the host (vbnmf_engine_step: one call and one read-back of lkh + statistics per iteration, SURVEY.md
    section 8(d)'s literal metric) -> `value_host_stepped` (top level) and the `host_stepped` block.
`warmup_effective` = W + the settle steps below; `setup` = seconds of untimed set-up (ingestion of X, engine creation =
the two tiled layouts + their upload, initial state + priming sweep, a second engine on the same matrix).
The roofline figures of k_sweep come from HIP events around its launches, on the engine's stream, in the host-stepped
pass (mean per repeat, median over the five repeats, all five in `kernel_ms_repeats`).  `roofline.traffic` is measured in the
same run at N = 1: rank 0 starts two short CHILD runs of this file under `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE; each
its own pass, no trace domain beside --pmc) and takes 2 x FETCH_SIZE + WRITE_SIZE per k_sweep launch; if rocprofv3 is missing
or a pass fails it falls back to the figure kept under profiles/ (`traffic_source` says which it was).

CPU references timed in the same run, on rank 0 at N = 1: `cpu_baseline` = the oracle's stored-entries step with
OpenMP on the box's cores (kind "port"); `cpu_reference_literal` = the oracle's LITERAL restatement of
src/vbnmf_update.cpp (dense, same operation order, ONE thread -- the reference builds without OpenMP, src/Makevars:1-2)
at a stated down-scale of the same matrix, with ns per matrix element and the extrapolation to the full size.

After the W warm-up steps the engine runs untimed settle steps (to ~0.3 s of load in all, `config.settle_steps`) so that
a short K measures the chip at its loaded clocks, not on its way up from idle.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode restarts|cells] [--small]

Beside the headline, at every N: `rank_sweep` = BASELINE config C4 (ranks 2..20 on the same matrix through
vb_factorize_sharded, reference defaults: wall seconds of the call as the library runs it -- four units in flight per
process --, per-rank iterations, and from a second call with ONE unit at a time the stepping / setup split per process;
that second call finds the sweep's layouts already cut); at N > 1
also `cells_partitioned` = config C5 (one factorisation, cells partitioned N-way, the library's all-reduce inside the
device-driven loop) with its per-GPU roofline and `allreduce_ms` (events around the collective); at N = 1 also
`restarts_small_matrix` = the reference's own size class (1030 x 450): sixteen restarts of a rank one loop at a time against
all sixteen stepped by one launch (vbnmf_batch_run), aggregate iterations per second; and `rank_sweep_nrun1` inside it: the
reference's default call shape, `vb_factorize(ranks = 2..9, nrun = 1)`, one loop at a time against one batch over the ranks.

N > 1 (one process per GPU under torch.distributed.run; a BARE `python bench.py --gpus N` starts those ranks itself as a
child process -- before torch is imported or the GPU touched -- relays rank 0's line and exits with their status):
  --mode restarts (default)  every GPU runs an independent restart of the same factorisation
                             (the reference's own parallelism: mpi.applyLB over runs, reference
                             R/bayesian.R:263); no data-path collective; weak scaling;
                             value = N * K / time.
  --mode cells               one factorisation, cells partitioned over the GPUs, one RCCL
                             all-reduce of [sw | rowSums(eh) | scalars] per step; strong scaling.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector peak (same guide); the sweep's arithmetic runs on the vector units
HYPER = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}


def make_workload(small: bool, world: int = 1, local_rank: int = 0):
    from ccfindr_amd import synth
    if world > 1:                                    # one generation per node (node_shared_matrix)
        name, r = (("C3-small 2k x 5k sparse, rank 10", 10) if small else ("C3 20k x 50k CSR-sparse (~5% nnz), rank 10", 10))
        return name, node_shared_matrix("c3s" if small else "c3", lambda: make_workload(small)[1], world, local_rank), r
    if small:
        n, m, r, k = 2000, 5000, 10, 5
        depth = None
        X = synth.simulate_data(n, [m // k] * k, alpha0=0.1, seed=3, nfactor=1)
        name = "C3-small 2k x 5k sparse, rank 10"
    else:
        n, m, r, k = 20000, 50000, 10, 10
        rng = np.random.default_rng(3)
        depth = np.round(rng.lognormal(np.log(1500.0), 0.3, size=m)).astype(np.int64)
        X = synth.simulate_data(n, [m // k] * k, alpha0=0.065, seed=3, depth=depth)
        name = "C3 20k x 50k CSR-sparse (~5% nnz), rank 10"
    X = synth.fill_empty(X, seed=3)
    return name, X, r


def node_rank():
    """This process's rank INSIDE its node (torch.distributed.run's LOCAL_RANK) -- not the device index, which a rehearsal
    may force to 0 for every rank (BENCH_ONE_DEVICE)."""
    try:
        return int(os.environ.get("LOCAL_RANK", "0"))
    except ValueError:
        return 0


def node_shared_matrix(tag, make, world, local_rank=None):
    """The synthetic matrix `make()` builds, ONCE PER NODE: under `--gpus N` the node's local rank 0 generates it (22 s for
    the headline matrix, two minutes for C5 on one core -- N ranks used to repeat that side by side) and puts its compressed
    columns into the node's memory file system; the other ranks map them.  Every rank gets a scipy CSC matrix over the
    same bytes.  BENCH_GEN_COUNTER names a file that receives one line per generation (tests/test_bench_launch.py)."""
    import scipy.sparse as sp

    def generate():
        X = make().tocsc()
        X.sort_indices()
        counter = os.environ.get("BENCH_GEN_COUNTER")
        if counter:
            with open(counter, "a") as fh:
                fh.write(f"{tag} pid {os.getpid()}\n")
        return X

    if world <= 1:
        return generate()
    local_rank = node_rank()
    import torch.distributed as dist
    from ccfindr_amd import node as shm
    box = [f"{os.getpid()}_{time.time_ns()}" if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)                       # one name for this run on every node
    base = os.path.join(shm.shm_dir(), f"vbnmf_bench_{tag}_{box[0]}")
    names = {k: f"{base}_{k}.npy" for k in ("indptr", "indices", "data", "shape")}
    X = None
    shared_ok = True
    if local_rank == 0:
        X = generate()
        # room in the node's memory file system for the three arrays (a container may give /dev/shm 64 MB: writing past a full
        # tmpfs fails) -- decided before anything is written
        need = X.nnz * 12 + 8 * (X.shape[1] + 1) + 4096
        shared_ok = shm.free_bytes() > 1.1 * need
    votes = [None] * world
    dist.all_gather_object(votes, bool(shared_ok))               # every rank learns every node's answer: one decision for the run
    if not all(votes):
        if dist.get_rank() == 0:
            print(f"bench: {tag}: no room in {shm.shm_dir()} for the shared matrix; every rank generates its own copy", file=sys.stderr)
        return X if X is not None else generate()
    if local_rank == 0:
        idt = np.int32 if X.nnz < 2 ** 31 - 1 else np.int64
        np.save(names["indptr"], X.indptr.astype(idt, copy=False))
        np.save(names["indices"], X.indices.astype(idt, copy=False))
        np.save(names["data"], np.asarray(X.data, dtype=np.float64))
        np.save(names["shape"], np.asarray(X.shape, dtype=np.int64))
    dist.barrier()
    if local_rank != 0:
        shape = tuple(int(v) for v in np.load(names["shape"]))
        X = sp.csc_matrix((np.load(names["data"], mmap_mode="r"), np.load(names["indices"], mmap_mode="r"),
                           np.load(names["indptr"], mmap_mode="r")), shape=shape, copy=False)
    dist.barrier()                                               # mapped everywhere: the names can go
    if local_rank == 0:
        for path in names.values():
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
    return X


def local_world():
    """Ranks of this node (torch.distributed.run exports it): the host threads of ingestion and layout cuts are shared."""
    try:
        return max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        return 1


def algorithmic_bytes(n, m, r, nnz):
    """canonical CSR fp64 value + int32 index, fp64 factors."""
    x_pass = 12 * nnz + 4 * (n + 1)
    bytes_iter = x_pass + 48 * (n * r + r * m)
    sweep = x_pass + 16 * (n * r + r * m)        # read lw, lh once, write the sw, sh statistics once
    return bytes_iter, sweep


def usable_cores():
    """Host cores this process may really use: affinity mask, cgroup quota, and the GPU box's
    stated share of 16 cores per GPU (asking OpenMP for all 128 visible threads oversubscribes it)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(X, r, wh0, nsteps):
    """The oracle's stored-entries restatement (OpenMP) on the same workload."""
    from oracle import vbnmf_oracle as O
    n, m = X.shape
    cores = min(usable_cores(), int(O.lib().oracle_max_threads()))
    p, i, x = X.indptr, X.indices, X.data
    wh = wh0
    lk = []
    t0 = time.perf_counter()
    for _ in range(nsteps):
        wh = O.update_csc(n, m, p, i, x, wh, HYPER, nthreads=cores)
        lk.append(wh["lkh"])
    dt = time.perf_counter() - t0
    return {"value": nsteps / dt, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"{nsteps} full steps of the same workload (oracle update_csc, OpenMP x{cores})"}, lk


def cpu_reference_literal(X, r, wh0, cells=2000, reps=2):
    """The reference's own CPU form: the oracle's literal dense restatement of src/vbnmf_update.cpp:19-101 on ONE
    thread, timed on the first `cells` cells of the workload (the dense form needs ~7 n*m temporaries: the full
    20k x 50k matrix would take 56 GB and about a minute per step) and extrapolated per matrix element."""
    from oracle import vbnmf_oracle as O
    n, m = X.shape
    cells = min(cells, m)
    A = np.asfortranarray(X.tocsc()[:, :cells].toarray())
    wh = {"lw": wh0["lw"], "lh": np.asfortranarray(wh0["lh"][:, :cells]), "eh": np.asfortranarray(wh0["eh"][:, :cells]),
          "ew": wh0["ew"]}
    t0 = time.perf_counter()
    for _ in range(reps):
        wh = O.update_dense(A, wh, HYPER)
    dt = (time.perf_counter() - t0) / reps
    return {"value": (1.0 / dt) * cells / m, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"{reps} steps of the literal dense restatement (oracle update_dense, 1 thread) on the first {cells} cells "
                      f"of the same matrix ({n} x {cells}); value = measured rate x {cells}/{m} (cost is linear in n*m)",
            "measured_it_per_s_at_sample": 1.0 / dt, "ns_per_matrix_element": 1e9 * dt / (n * cells),
            "extrapolated_s_per_iteration_full_size": dt * m / cells}


def make_c5(small):
    from ccfindr_amd import synth
    if small:
        n, m, r, k, mean_depth = 3000, 16000, 20, 8, 200.0
    else:
        n, m, r, k, mean_depth = 30000, 200000, 20, 20, 1950.0
    depth = np.round(np.random.default_rng(5).lognormal(np.log(mean_depth), 0.3, size=m)).astype(np.int64)
    X = synth.fill_empty(synth.simulate_data(n, [m // k] * k, alpha0=0.1, seed=5, depth=depth), seed=5)
    return X, n, m, r


def cells_partitioned_sample(world, rank, local_rank, barrier, steps, small=False, check=True):
    """BASELINE config C5 as a side measurement of an N > 1 run: ONE factorisation of a 30 000 x 200 000 (~5 % stored)
    matrix at rank 20, cells partitioned over the N GPUs, the per-step all-reduce of [sw | rowSums(eh) | scalars] issued
    by the library (RCCL over xGMI) inside the device-driven loop; strong scaling.  Beside the loop's rate: the per-GPU
    roofline of a step (algorithmic bytes and flops of this GPU's partition over the step time, both fractions) and
    `allreduce_ms`, the all-reduce alone (HIP events either side of vbnmf_engine_allreduce on the engine's stream, in a
    host-stepped pass: in the loop it travels beside the cell-side sweep).  Rank 0 checks the first step's evidence
    against the CPU oracle."""
    import torch
    import ccfindr_amd as C
    from ccfindr_amd import synth
    from ccfindr_amd.parallel import CellPartitionedEngine
    if small:
        n, m, r = 3000, 16000, 20
    else:
        n, m, r = 30000, 200000, 20
    X = node_shared_matrix("c5s" if small else "c5", lambda: make_c5(small)[0], world, local_rank)
    assert X.shape == (n, m)
    # every rank ingests ITS column block and nothing else (the engine of a partition never looks at other cells)
    from ccfindr_amd.parallel import cell_partition
    cb0, ce0 = cell_partition(m, world)[rank]
    M = C.CountMatrix(X[:, cb0:ce0] if world > 1 else X)
    eng = CellPartitionedEngine(M, r, device=local_rank, block=(cb0, ce0, m) if world > 1 else None)
    wh = synth.random_state(n, m, r, HYPER, seed=1005)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    device_loop = eng.native or world == 1
    first_lkh = None
    times = []
    if device_loop:
        first = eng.run(HYPER, Itmax=5, Tol=0.0, flags=(False,) * 4, history=True)      # warm-up; history[0] is checked below
        first_lkh = float(first["history"][0, 0])
        for _ in range(3):
            barrier()
            t0 = time.perf_counter()
            res = eng.run(HYPER, Itmax=steps, Tol=0.0, flags=(False,) * 4)
            barrier()
            times.append(time.perf_counter() - t0)
        lkh_last = res["lkh"]
    else:
        # a backend that cannot carry the library's collective (gloo rehearsals): host-stepped, exchange staged by torch
        first_lkh, _ = eng.step(HYPER)
        for _ in range(4):
            eng.step(HYPER)
        for _ in range(3):
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                lkh_last, _ = eng.step(HYPER)
            barrier()
            times.append(time.perf_counter() - t0)
    # the all-reduce alone: events on the engine's stream either side of the collective, host-stepped
    ar_ms = None
    if eng.native:
        ctx = eng.engine.stream_context
        pairs = []
        for _ in range(12):
            eng.engine.step_local(HYPER)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with ctx():
                e0.record()
            eng.engine.allreduce()
            with ctx():
                e1.record()
            eng.engine.step_finish()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        ar_ms = float(np.median([a.elapsed_time(b) for a, b in pairs[2:]]))
    cb, ce = eng.cols
    S = X
    nnz_local = int(S.indptr[ce] - S.indptr[cb])
    m_local = ce - cb
    bytes_gpu = 12 * nnz_local + 4 * (n + 1) + 48 * (n * r + r * m_local)       # SURVEY section 8(d), this GPU's share
    flops_gpu = 10 * r * nnz_local
    out = {"workload": f"C5{'-small' if small else ''} {n} x {m} (~5 % stored, nnz {int(X.nnz)}), rank {r}, cells partitioned {world}-way",
           "loop": ("device-driven (vbnmf_engine_run + vbnmf_comm: ncclAllReduce enqueued from C++ beside the cell-side sweep)"
                    if device_loop else "host-stepped, exchange staged through torch.distributed (rehearsal backend)"),
           "collective": ("library communicator: " + (os.environ.get("VBNMF_RCCL_LIB") and "stand-in named by VBNMF_RCCL_LIB (rehearsal)" or "RCCL"))
                         if eng.native else "torch.distributed",
           "scaling": "strong", "steps": steps, "repeats_ms_per_step": [1e3 * t / steps for t in times],
           "allreduce_bytes_per_step": 8 * (n * r + r + 2) + 8 * 1025,      # [sw | rowSums(eh) | 2] beside the sweep + the evidence slots behind it
            "allreduce_ms": ar_ms, "lkh_last": lkh_last,
           "per_gpu": {"cells": m_local, "nnz": nnz_local, "algorithmic_bytes_per_step": bytes_gpu, "flops_per_step": flops_gpu}}
    eng.close()
    if rank == 0 and check:
        from oracle import vbnmf_oracle as O
        ref = O.update_csc(n, m, S.indptr, S.indices, S.data, wh, HYPER, nthreads=usable_cores())
        out["lkh_rel_err_first_step_vs_cpu_oracle"] = abs(first_lkh / ref["lkh"] - 1)
    M.close()
    return out, times


def rank_sweep_sample(M, world, rank, local_rank, barrier, small=False):
    """BASELINE config C4: vb_factorize over ranks 2..20 on the headline matrix with the reference's defaults (hyper-parameter
    updates on, Tol 1e-5), the (run, rank) units dealt longest-first over the N processes (one per GPU) -- no data-path
    collective (reference R/bayesian.R:261-263, 316); one ingestion and one pair of layouts per node.  Wall seconds of the
    call (slowest process), per-rank iterations, and per process the split into layouts / units / exchange and into device
    stepping against everything else."""
    import torch.distributed as dist
    from ccfindr_amd import parallel
    ranks = list(range(2, 7)) if small else list(range(2, 21))

    def one_call(concurrent):
        tm = {}
        barrier()
        t0 = time.perf_counter()
        res = parallel.vb_factorize_sharded(M, ranks=ranks, nrun=1, Itmax=2000, Tol=1e-5, seed=7, device=local_rank, timings=tm,
                                            concurrent=concurrent)
        t_call = time.perf_counter() - t0
        barrier()
        t_wall = time.perf_counter() - t0
        units = tm.get("unit_detail") or []
        mine = {"process": rank, "call_s": t_call, "layouts_s": tm["layout_s"], "units_s": tm["units_s"], "exchange_s": tm["gather_s"],
                "stepping_s": sum(u["loop_s"] for u in units), "ranks": [u["rank"] for u in units]}
        mine["setup_s"] = t_call - mine["stepping_s"]
        rows = [mine]
        if world > 1:
            rows = [None] * world
            dist.all_gather_object(rows, mine)                   # (small records; outside the timed call)
        return res, t_wall, rows

    # the call as the library runs it (several units in flight per process: their host sides overlap other units' stepping,
    # so stepping and set-up cannot be told apart) ...
    res, wall, rows = one_call(4)
    # ... and once more with ONE unit at a time, for the split into device stepping and everything else
    res1, wall1, rows1 = one_call(1)
    assert list(res.nsteps) == list(res1.nsteps) and list(res.measure["lml"]) == list(res1.measure["lml"])
    return {"workload": f"ranks {ranks[0]}..{ranks[-1]} on the headline matrix through vb_factorize_sharded, reference defaults "
                        f"(hyper updates on, Tol 1e-5), {world} process(es)",
            "wall_s": wall, "units_in_flight_per_process": 4, "call_s_slowest_process": max(q["call_s"] for q in rows),
            "one_unit_at_a_time": {"wall_s": wall1, "stepping_s_total": sum(q["stepping_s"] for q in rows1),
                                   "setup_s_slowest_process": max(q["setup_s"] for q in rows1), "per_process": rows1},
            "iterations_by_rank": dict(zip([int(v) for v in res.ranks], [int(v) for v in res.nsteps])),
            "iterations_total": int(sum(res.nsteps)),
            "best_rank_by_lml": int(res.ranks[int(np.argmax(res.measure["lml"]))]),
            "per_process": [{k: q[k] for k in ("process", "call_s", "layouts_s", "units_s", "exchange_s", "ranks")} for q in rows]}


def small_matrix_restarts_sample():
    """Side measurement (N = 1, rank 0): the reference's OWN size class -- its shipped data set is 1030 x 450 -- where a step is
    latency bound and the parallelism is the nrun restarts of a rank (reference R/bayesian.R:260-261): 16 restarts of rank 5,
    400 iterations each, one loop at a time against all sixteen stepped by one launch (vbnmf_batch_run)."""
    import torch
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.drop_empty(synth.simulate_data(1030, (150, 150, 150), seed=3, sparse=True))
    M = C.CountMatrix(X)
    n, m = X.shape
    r, B, iters = 5, 16, 400
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    whs = [synth.random_state(n, m, r, hy, seed=b) for b in range(B)]
    out = {"workload": f"{n} x {m} counts ({X.nnz} stored), rank {r}, {B} restarts x {iters} iterations, hyper updates on", "dtype": "f64"}
    kw = dict(Itmax=iters, Tol=0.0, n0=10, dn=1)
    for label, grid in (("one_at_a_time", None), ("batched", C.batch_grid(B))):
        engs = [C.VBEngine(M, r, grid=grid) for _ in range(B)]
        for eng, wh in zip(engs, whs):
            eng.set_state(wh["lw"], wh["lh"], wh["eh"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if grid is None:
            res = [eng.run(hy, **kw) for eng in engs]
        else:
            res = C.run_batch(engs, [hy] * B, **kw)
        dt = time.perf_counter() - t0
        assert all(o["it"] == iters for o in res)
        out[label] = {"iterations_per_s": B * iters / dt, "seconds": dt, "lkh_first": res[0]["lkh"]}
        for eng in engs:
            eng.close()
    out["speedup"] = out["batched"]["iterations_per_s"] / out["one_at_a_time"]["iterations_per_s"]
    out["lkh_rel_diff"] = abs(out["batched"]["lkh_first"] / out["one_at_a_time"]["lkh_first"] - 1)
    # the reference's DEFAULT call has nrun = 1: then the units are the ranks of the sweep (engines made one row width wide share a batch)
    kw = dict(ranks=range(2, 10), nrun=1, verbose=0, Tol=0.0, seed=5, Itmax=iters, unif_stop=False)
    C.vb_factorize(M, **kw)                                       # (layouts of this geometry cut and cached)
    sweep = {"workload": f"vb_factorize(ranks = 2..9, nrun = 1), {iters} iterations per rank, on the same matrix"}
    for label, extra in (("one_at_a_time_s", dict(batch=1)), ("batched_s", {})):
        t0 = time.perf_counter()
        C.vb_factorize(M, **kw, **extra)
        sweep[label] = time.perf_counter() - t0
    sweep["speedup"] = sweep["one_at_a_time_s"] / sweep["batched_s"]
    out["rank_sweep_nrun1"] = sweep
    M.close()
    return out


def measure_sweep_traffic(timeout_s=300.0):
    """HBM bytes per k_sweep launch measured IN THIS RUN, on this box: two child runs of this file under
    `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE: the TCC block cannot hold both in one pass; no trace domain beside
    --pmc), a short headline-only workload each; bytes = 2 x FETCH_SIZE + WRITE_SIZE in KB -- the gfx950 correction of
    /opt/skills/guides/MI355X_MICROARCH.md for a wide coalesced streaming read.  Returns (bytes, detail) or (None, reason);
    never raises: a box without rocprofv3, a counter pass that fails or takes too long leaves the figure to the file under
    profiles/.  BENCH_NO_TRAFFIC=1 skips it."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if os.environ.get("BENCH_NO_TRAFFIC"):
        return None, "skipped (BENCH_NO_TRAFFIC)"
    # Already under a profiler (rocprofv3 -- python3 bench.py ...): a nested launcher would inherit the outer profiler's
    # preloaded tool library, which touches the GPU before the inner launcher execs its target -- the exec this pool forbids --
    # and the extra passes would skew the traced run (profiles/ubench/r04/stage_ids_ab.sh ran exactly that).
    prof = ("ROCPROF", "ROCP_", "ROCPROFILER", "ROCTRACER")
    marks = [k for k in os.environ if k.startswith(prof)]
    preload = os.environ.get("LD_PRELOAD", "")
    if marks or "rocprof" in preload or "roctx" in preload:
        return None, "already under a profiler (%s)" % ", ".join(marks[:3] or ["LD_PRELOAD"])
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not tool:
        return None, "rocprofv3 not found"
    means = {}
    t_all = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        env = {k: v for k, v in os.environ.items() if not k.startswith(prof) and k != "LD_PRELOAD"}
        env.update(BENCH_NO_SWEEP="1", BENCH_NO_TRAFFIC="1", BENCH_NO_SMALL="1", TMPDIR="/tmp")      # (the headline's sweeps only)
        cmd = [tool, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable,
               os.path.abspath(__file__), "--steps", "24", "--warmup", "2", "--no-cpu", "--no-ml"]
        try:
            proc = subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=timeout_s / 2)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)          # exactly the group this call started
                proc.wait()
                shutil.rmtree(out, ignore_errors=True)
                return None, f"the {counter} pass exceeded {timeout_s / 2:.0f} s"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                shutil.rmtree(out, ignore_errors=True)
                return None, f"the {counter} pass ended with status {rc}"
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                    if r["Counter_Name"] == counter and "k_sweep<" in r["Kernel_Name"]]
            shutil.rmtree(out, ignore_errors=True)
            if not vals:
                return None, f"no k_sweep dispatch in the {counter} pass"
            means[counter] = (sum(vals) / len(vals), len(vals))
        except Exception as exc:                              # noqa: BLE001 -- the figure is optional, the bench line is not
            shutil.rmtree(out, ignore_errors=True)
            return None, f"{type(exc).__name__}: {exc}"
    f, w = means["FETCH_SIZE"][0], means["WRITE_SIZE"][0]
    return (2.0 * f + w) * 1024.0, {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
                                    "dispatches": means["FETCH_SIZE"][1], "seconds": time.perf_counter() - t_all}


def timed_repeats(fn, barrier, repeats=5, after=None):
    """`repeats` x [barrier, fn(), barrier] -> list of seconds; `after()` runs outside the timed region of every repeat."""
    out = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        fn()
        barrier()
        out.append(time.perf_counter() - t0)
        if after is not None:
            after()
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run as a child process."""
    import socket
    import subprocess
    with socket.socket() as sk:                      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=("restarts", "cells"), default="restarts")
    ap.add_argument("--small", action="store_true", help="2k x 5k smoke-sized workload (not the headline)")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-ml", action="store_true", help="skip the ML-NMF side measurement")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--rank", type=int, default=0, help="diagnostic: another rank on the same matrix (the metric is rank 10)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Bare launch (`python bench.py --gpus N`): this process has not imported torch or touched the GPU, so it starts
        # the N ranks as CHILD processes (one per GPU, torch.distributed.run on 127.0.0.1), relays their output -- rank 0
        # prints the JSON line -- and leaves with their status.  Never an exec of a process that holds the GPU.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch
    import torch.distributed as dist
    import ccfindr_amd as C
    from ccfindr_amd import synth

    # BENCH_BACKEND=gloo rehearses the multi-process paths with several ranks on ONE GPU (RCCL refuses two ranks on
    # a device); BENCH_ONE_DEVICE=1 then puts every rank on cuda:0.  The driver's runs use neither.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1 and local_world() > 1 and not os.environ.get("VBNMF_HOST_THREADS"):
        # the node's ranks ingest and cut side by side: each takes its share of the host's cores (default: up to 32 each)
        from ccfindr_amd.engine import set_host_threads
        from ccfindr_amd.node import usable_cores as node_cores
        set_host_threads(max(1, min(32, node_cores() // local_world())))       # (affinity mask and cgroup quota)
    name, X, r = make_workload(args.small, world, local_rank)
    if args.rank > 0 and args.rank != r:
        r = args.rank
        name = name.replace("rank 10", f"rank {r} (diagnostic, not the headline rank)")
    n, m = X.shape
    nnz = int(X.nnz)
    setup = {}                                   # untimed set-up, reported beside the metric (never part of `value`)
    t_s = time.perf_counter()
    M = C.CountMatrix(X)
    setup["ingest_s"] = time.perf_counter() - t_s
    t_s = time.perf_counter()
    from ccfindr_amd.engine import device_warmup
    device_warmup(local_rank)           # this process's first use of the device (HIP context, first stream): 0.13 s that
    setup["device_init_s"] = time.perf_counter() - t_s          # are not the engine's -- they used to hide in engine_create_s

    if args.mode == "cells" and world > 1:
        from ccfindr_amd.parallel import CellPartitionedEngine
        eng = CellPartitionedEngine(M, r, device=local_rank)
        wh0 = synth.random_state(n, m, r, HYPER, seed=1003)
        eng.set_state(wh0["lw"], wh0["lh"], wh0["eh"])
        step = lambda: eng.step(HYPER)
        units_per_step = 1
        scaling = "strong"
    else:
        t_s = time.perf_counter()
        eng = C.VBEngine(M, r, device=local_rank)
        setup["engine_create_s"] = time.perf_counter() - t_s          # tiled layouts of both sides + their upload
        wh0 = synth.random_state(n, m, r, HYPER, seed=1003 + rank)
        t_s = time.perf_counter()
        eng.set_state(wh0["lw"], wh0["lh"], wh0["eh"])
        setup["set_state_s"] = time.perf_counter() - t_s              # state upload + the priming sweep
        t_s = time.perf_counter()
        C.VBEngine(M, r, device=local_rank).close()
        setup["engine_create_cached_s"] = time.perf_counter() - t_s   # a second engine on the same matrix and geometry
        step = lambda: eng.step(HYPER)
        units_per_step = world
        scaling = "weak"

    # evidence of the first steps, kept for the match against the CPU restatement
    ncheck = 0 if args.no_cpu else max(1, args.cpu_steps)
    gpu_lk = []
    for _ in range(args.warmup):
        lkh, _ = step()
        gpu_lk.append(lkh)

    # Clock settle: the driver may ask for a very short run (K = 20, W = 5 is ~7 ms of GPU work), and the chip needs tens
    # of milliseconds of load to leave its idle clocks (k_sweep reads 215 us in the first 5 ms after start-up and 179 us
    # 25 ms later).  After the W warm-up steps, untimed steps are added until ~0.3 s of stepping lie behind the engine.
    settle = max(0, 1200 - args.warmup)
    if hasattr(eng, "run") and getattr(eng, "native", True):
        eng.run(HYPER, Itmax=settle, Tol=0.0, n0=10, dn=1, flags=(False,) * 4)
    else:
        for _ in range(settle):
            step()

    base_eng = getattr(eng, "engine", eng)
    REPEATS = 5
    last = {}

    def host_pass():
        for _ in range(args.steps):
            last["lkh"], _ = step()

    base_eng.timing_enable(True)
    sweep_repeats = []                           # mean k_sweep time (HIP events on the engine's stream) of every repeat
    host_times = timed_repeats(host_pass, barrier, REPEATS, after=lambda: sweep_repeats.append(base_eng.timing_get()))
    dt = float(np.median(host_times))
    lkh = last["lkh"]
    per_repeat_ms = [ms / cnt for ms, cnt in sweep_repeats if cnt]
    sweep_cnt = sum(cnt for _, cnt in sweep_repeats)
    sweep_ms = (float(np.median(per_repeat_ms)) * sweep_cnt) if per_repeat_ms else 0.0      # median repeat, as for `value`

    # The same K steps with the loop driven by the device (vbnmf_engine_run: the product's default path,
    # ccfindr_amd/bayesian.py::vb_run_rank): hyper_update and the stopping rule of vb_iterate evaluated on the GPU,
    # steps queued ahead, lkh + the four statistics of EVERY step still delivered to the host through the history.
    # This is `value`; the host-stepped rate of the loop above is reported beside it.
    dt_dev = None
    dev_times = None
    if hasattr(eng, "run") and getattr(eng, "native", True):
        base_eng.timing_enable(False)

        def dev_pass():
            last["res"] = eng.run(HYPER, Itmax=args.steps, Tol=0.0, n0=10, dn=1, flags=(False,) * 4, history=True)
            if last["res"]["it"] != args.steps:
                raise SystemExit(f"device loop ran {last['res']['it']} steps, expected {args.steps}")

        dev_times = timed_repeats(dev_pass, barrier, REPEATS)
        dt_dev = float(np.median(dev_times))
        lkh_dev = last["res"]["lkh"]
        # one more pass with the reference's default hyper.update = TRUE (n0 = 10, dn = 1), to
        # show what the Newton updates of aw, ah on the device cost per step (reported, not `value`)
        hyper_on = None
        if world == 1:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            res_h = eng.run(HYPER, Itmax=args.steps, Tol=0.0, n0=10, dn=1, flags=(True,) * 4)
            torch.cuda.synchronize()
            dt_h = time.perf_counter() - t1
            hyper_on = {"value": res_h["it"] / dt_h, "unit": "iterations/s", "ms_per_step": 1e3 * dt_h / max(res_h["it"], 1),
                        "steps": res_h["it"], "hyper_last": res_h["hyper"]}

    # The maximum-likelihood NMF step of factorize() (reference R/factorize.R:2-27 + :40-49)
    # on the same matrix and rank: K steps host-stepped (sweep timing) and K device-driven, reported beside the headline
    # (never part of `value`).
    ml = None
    if world == 1 and hasattr(eng, "ml_step") and not args.no_ml:
        rng = np.random.default_rng(2003)
        w_ml, h_ml = rng.uniform(size=(n, r)), rng.uniform(size=(r, m))
        eng.ml_set_state(w_ml, h_ml)
        ml_lk = [eng.ml_step() for _ in range(max(args.warmup, 2))]
        base_eng.timing_enable(True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            lk_ml = eng.ml_step()
        torch.cuda.synchronize()
        dt_ml = time.perf_counter() - t2
        ml_ms, ml_cnt = base_eng.timing_get()
        base_eng.timing_enable(False)
        x_pass = 12 * nnz + 4 * (n + 1)
        ml_bytes = 2 * x_pass + 24 * (n * r + r * m)       # two passes over X (H then W), factors read twice, statistics written once
        # the same K steps driven by the device (vbnmf_engine_ml_run, factorize()'s default path)
        eng.ml_set_state(w_ml, h_ml)
        for _ in range(max(args.warmup, 2)):
            eng.ml_step()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        run_ml = eng.ml_run(Itmax=args.steps, Tol=0.0)
        torch.cuda.synchronize()
        dt_ml_dev = time.perf_counter() - t4
        ml = {"value": run_ml["it"] / dt_ml_dev, "unit": "iterations/s", "ms_per_step": 1e3 * dt_ml_dev / max(run_ml["it"], 1),
              "steps": run_ml["it"], "loop": "device-driven (vbnmf_engine_ml_run)",
              "host_stepped": {"value": args.steps / dt_ml, "ms_per_step": 1e3 * dt_ml / args.steps},
              "lk_last": run_ml["lk"], "sweeps_ms_per_step": 2.0 * ml_ms / max(ml_cnt, 1),
              "roofline": {"bound": "hbm", "kernel": "k_sweep1 x2", "algorithmic_bytes_per_step": ml_bytes,
                           "achieved": ml_bytes / (2.0 * ml_ms / max(ml_cnt, 1) * 1e-3) / 1e9 if ml_cnt else None,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ml_bytes / (2.0 * ml_ms / max(ml_cnt, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS if ml_cnt else None}}
        if not args.no_cpu:
            from oracle import mlnmf_oracle as OM
            cores = usable_cores()
            S = X.tocsc()
            t3 = time.perf_counter()
            w_c, h_c, cpu_lk = w_ml, h_ml, []
            for _ in range(2):
                o = OM.update_csc(n, m, S.indptr, S.indices, S.data, w_c, h_c, nthreads=cores)
                w_c, h_c = o["ew"], o["eh"]
                cpu_lk.append(o["lk"])
            ml["cpu_baseline"] = {"value": 2 / (time.perf_counter() - t3), "unit": "iterations/s", "cores": cores, "kind": "port",
                                  "sample": f"2 full steps of the same workload (oracle mlnmf update_csc, OpenMP x{cores})"}
            ml["lk_rel_err_first_steps"] = max(abs(g / c - 1) for g, c in zip(ml_lk[:2], cpu_lk))

    if world > 1:
        # MAX over ranks of every repeat, then the median of those
        t = torch.tensor([host_times, dev_times if dev_times is not None else [0.0] * REPEATS], dtype=torch.float64,
                         device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        host_times = [float(v) for v in t[0].tolist()]
        dt = float(np.median(host_times))
        if dev_times is not None:
            dev_times = [float(v) for v in t[1].tolist()]
            dt_dev = float(np.median(dev_times))

    if rank == 0:
        bytes_iter, bytes_sweep = algorithmic_bytes(n, m, r, nnz)
        host_value = units_per_step * args.steps / dt
        host_ms = 1e3 * dt / args.steps
        if dt_dev is not None:
            value, ms_step, loop = units_per_step * args.steps / dt_dev, 1e3 * dt_dev / args.steps, "device-driven (vbnmf_engine_run)"
        else:
            value, ms_step, loop = host_value, host_ms, "host-stepped (vbnmf_engine_step)"
        sweep_s = (sweep_ms / max(sweep_cnt, 1)) * 1e-3
        achieved = bytes_sweep / sweep_s / 1e9 if sweep_cnt else None
        info = base_eng.layout_info()
        traffic, traffic_source, traffic_detail = None, None, None
        if world == 1 and not args.small and args.rank == 0 and not args.no_traffic:
            # measured in this run, on this box: two short child runs under rocprofv3 --pmc (the engine of THIS process is idle meanwhile)
            traffic, traffic_detail = measure_sweep_traffic()
            if traffic is not None:
                traffic_source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE over two child runs of this "
                                  "command (24 steps each), 2 x FETCH_SIZE + WRITE_SIZE per k_sweep launch")
        for tname in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):          # newest rocprofv3 --pmc summary kept under profiles/
            tpath = os.path.join(ROOT, "profiles", tname)
            if traffic is None and os.path.exists(tpath) and not args.small and args.rank in (0, 10):
                try:
                    traffic = json.load(open(tpath)).get("sweep_hbm_bytes_per_launch")
                    traffic_source = (f"profiles/{tname} (rocprofv3 --pmc passes of an earlier run of this command; not measured in this run"
                                      + (f": {traffic_detail}" if isinstance(traffic_detail, str) else "") + ")")
                    break
                except Exception:
                    traffic = None
        out = {
            "metric": "VB-NMF update iterations/sec (20k x 50k sparse counts, rank 10)",
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "value_host_stepped": host_value, "warmup_effective": args.warmup + settle,
            "setup": dict(setup, note="seconds, outside every timed region: ingestion of X, this process's first use of the device, engine creation (tiled layouts "
                                      "cut on the host + upload), initial state + priming sweep"),
            "config": {"workload": name, "n_genes": n, "n_cells": m, "nnz": nnz, "rank": r,
                       "mode": args.mode if world > 1 else "single", "hyper": "fixed aw=bw=ah=bh=1",
                       "loop": loop, "lkh_last": lkh_dev if dt_dev is not None else lkh,
                       "timing": f"median of {REPEATS} repeats of the {args.steps}-step region",
                       "settle_steps": settle},
            "roofline": {"bound": "hbm", "kernel": "k_sweep", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "traffic_counters": traffic_detail if isinstance(traffic_detail, dict) else None,
                         "algorithmic_bytes_per_launch": bytes_sweep, "kernel_ms": sweep_s * 1e3,
                         "kernel_ms_repeats": per_repeat_ms,
                         "streamed_bytes_per_launch": info["stream_bytes_per_step"]},
            # SURVEY.md section 8(d) asks for both fractions: algorithmic flops = 10 r per stored entry
            "roofline_fp64": {"bound": "fp64-valu", "kernel": "k_sweep", "flops_per_launch": 10 * r * nnz,
                              "achieved": (10 * r * nnz / sweep_s / 1e12) if sweep_cnt else None, "peak": FP64_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": (10 * r * nnz / sweep_s / 1e12 / FP64_PEAK_TFLOPS) if sweep_cnt else None},
            "iteration_roofline": {"bytes_iter": bytes_iter, "achieved_GBs": bytes_iter * (value / units_per_step) / 1e9,
                                   "frac": bytes_iter * (value / units_per_step) / 1e9 / HBM_PEAK_GBS},
        }
        out["repeats_ms_per_step"] = [1e3 * t / args.steps for t in (dev_times if dev_times is not None else host_times)]
        out["host_stepped"] = {"value": host_value, "unit": "iterations/s", "ms_per_step": host_ms, "steps": args.steps,
                               "lkh_last": lkh, "repeats_ms_per_step": [1e3 * t / args.steps for t in host_times]}
        if dt_dev is not None and hyper_on:
            out["hyper_updates_on"] = hyper_on
        if ml:
            out["ml_nmf"] = ml
        if not args.no_cpu and world == 1:
            wh_cpu = synth.random_state(n, m, r, HYPER, seed=1003)
            cb, cpu_lk = cpu_baseline(X, r, wh_cpu, ncheck)
            out["cpu_baseline"] = cb
            out["cpu_reference_literal"] = cpu_reference_literal(X, r, wh_cpu)
            k = min(len(cpu_lk), len(gpu_lk))
            if k:
                out["elbo_rel_err_first_steps"] = max(abs(g / c - 1) for g, c in zip(gpu_lk[:k], cpu_lk[:k]))
    # Everything below is a side measurement.  At N > 1 it involves collectives (tiny ones in the sweep's control plane, the
    # library's all-reduce in the partitioned sample): it can only hang where a peer or the fabric does, so the headline
    # (measured above, without any collective) is safe behind a watchdog that prints it, records the hang and leaves.
    dog = None
    if world > 1:
        import threading

        def give_up():
            if rank == 0:
                for key in ("rank_sweep", "cells_partitioned"):
                    out.setdefault(key, {"error": "timed out (watchdog 600 s): a collective of this side measurement hung", "hang": True})
                print(json.dumps(out), flush=True)
            os._exit(3)

        dog = threading.Timer(600.0, give_up)
        dog.daemon = True
        dog.start()
    # Config C4 beside the headline, at every N: the rank sweep through vb_factorize_sharded (no collective on its data path).
    rank_sweep = None
    if args.mode == "restarts" and args.rank == 0 and not os.environ.get("BENCH_NO_SWEEP"):
        try:
            eng.close()
            rank_sweep = rank_sweep_sample(M, world, rank, local_rank, barrier, small=args.small)
        except Exception as exc:                                   # noqa: BLE001 -- the headline must still be printed
            rank_sweep = {"error": f"{type(exc).__name__}: {exc}"}
        if rank == 0:
            out["rank_sweep"] = rank_sweep
    if world == 1 and args.mode == "restarts" and args.rank == 0 and not args.small and not os.environ.get("BENCH_NO_SMALL"):
        try:
            out["restarts_small_matrix"] = small_matrix_restarts_sample()
        except Exception as exc:                                   # noqa: BLE001 -- the headline must still be printed
            out["restarts_small_matrix"] = {"error": f"{type(exc).__name__}: {exc}"}
    # N > 1, default mode: after the headline (independent restarts, no collective) one C5-shaped cell-partitioned
    # factorisation is run on the same N GPUs, so that the RCCL path of the library is measured on hardware too.  It can
    # only hang where the fabric does, so the headline line is safe behind a watchdog that prints it and leaves.
    if world > 1 and args.mode == "restarts" and not os.environ.get("BENCH_NO_CELLS"):
        try:
            eng.close()
            cp, times = cells_partitioned_sample(world, rank, local_rank, barrier, max(50, min(args.steps, 300)) if not args.small else 20,
                                                 small=args.small, check=not args.no_cpu)
            t = torch.tensor(times, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            med = float(np.median(t.tolist()))
            cp["value"] = cp["steps"] / med
            cp["unit"] = "iterations/s"
            cp["ms_per_step"] = 1e3 * med / cp["steps"]
            g = cp["per_gpu"]
            cp["roofline"] = {"bound": "fp64-valu (rank 20) / hbm", "per": "GPU and step (whole step, all-reduce included)",
                              "hbm": {"achieved": g["algorithmic_bytes_per_step"] / med * cp["steps"] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": g["algorithmic_bytes_per_step"] / med * cp["steps"] / 1e9 / HBM_PEAK_GBS},
                              "fp64": {"achieved": g["flops_per_step"] / med * cp["steps"] / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": g["flops_per_step"] / med * cp["steps"] / 1e12 / FP64_PEAK_TFLOPS}}
            if rank == 0:
                out["cells_partitioned"] = cp
        except Exception as exc:                                   # noqa: BLE001 -- the headline must still be printed
            if rank == 0:
                out["cells_partitioned"] = {"error": f"{type(exc).__name__}: {exc}"}
    if dog is not None:
        dog.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)

    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
