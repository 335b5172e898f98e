"""Multi-GPU forms of the VB-NMF path: one process per GPU, torch.distributed over RCCL/xGMI.

The reference's only parallelism is ``Rmpi::mpi.applyLB(seq_len(nrun), FUN=vb_iterate, bundle)``
(reference R/bayesian.R:262-263): independent restarts handed to MPI slaves, the whole
matrix shipped to each, no communication while iterating.  Two native equivalents:

* ``vb_factorize_sharded``  rank sweep / restarts: every (run, rank) factorisation is an
  independent unit; units are dealt to the GPUs longest-first (cost ~ rank); X is replicated;
  NO data-path collective -- the per-unit results (host objects) are gathered at the end.
* ``CellPartitionedEngine``  one factorisation with the cells (columns) split across GPUs.
  Gene-side state is replicated; per step each GPU leaves its partial gene statistics, its
  rowSums(eh) and four scalars in the engine's reduce buffer and ONE all-reduce (sum, fp64)
  makes them global.  The buffer is n*R + R + 4 doubles (4.8 MB at
  30k genes, rank 20): latency-bound on xGMI, so it is a single contiguous collective.
"""
from __future__ import annotations

import math

import numpy as np

from .bayesian import _close_engines, assemble_run, make_bundle, plan_geometry, select_best, vb_run_rank
from .engine import EPS, VBEngine


# ---------------------------------------------------------------------------------------
# rank sweep / restarts
# ---------------------------------------------------------------------------------------
class ShardedRunError(RuntimeError):
    """A (run, rank) unit of a sharded sweep failed on some process; raised on EVERY process."""


def lpt_schedule(costs, n_workers):
    """Longest-processing-time-first assignment.  Returns ``n_workers`` lists of task indices;
    ties go to the lowest worker id, so every process computes the same schedule."""
    order = sorted(range(len(costs)), key=lambda t: (-costs[t], t))
    loads = [0.0] * n_workers
    out = [[] for _ in range(n_workers)]
    for t in order:
        w = min(range(n_workers), key=lambda k: (loads[k], k))
        out[w].append(t)
        loads[w] += costs[t]
    return out


def sweep_tasks(ranks, nrun):
    """(run, rank) units of a rank sweep with nrun restarts, and their relative cost (~ rank:
    the sweep's flops per entry are 10*rank)."""
    tasks = [(irun, int(r)) for irun in range(1, nrun + 1) for r in ranks]
    return tasks, [float(r) for _, r in tasks]


# columns of the per-unit record that travels through the process group (one small fp64 tensor, summed)
_REC_DONE, _REC_LK0, _REC_AW, _REC_BW, _REC_AH, _REC_BH, _REC_NSTEPS, _REC_FAIL, _REC_OWNER, _REC_UNIF0 = range(10)


def _unit_offsets(tasks, n, m):
    """Byte offsets of every unit's four factor matrices [ew | eh | sdw | sdh] in the node's result segment."""
    off, table = 0, {}
    for t, (_, r) in enumerate(tasks):
        table[t] = off
        off += 8 * 2 * (n * r + r * m)
    return table, off


def _unit_views(seg, base, n, m, r):
    nr, rm = 8 * n * r, 8 * r * m
    return {"ew": seg.array(base, (n, r)), "eh": seg.array(base + nr, (r, m)),
            "dw": seg.array(base + nr + rm, (n, r)), "dh": seg.array(base + 2 * nr + rm, (r, m))}


def _any_failed(flag, world, group, device):
    """True on every process when `flag` is set on any: one all-reduce that every process of the group reaches (it is also
    a barrier).  A single process just returns its own flag."""
    if world <= 1:
        return bool(flag)
    import torch
    import torch.distributed as dist
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64)
    if dist.get_backend(group) == "nccl":
        t = t.to(torch.device("cuda", device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.cpu()[0]) > 0.0


def _remove_files(paths):
    """Unlink what this process put into the node's memory file system: finished segments, the `.part` a failed cut left,
    its failure note.  tmpfs pages are RAM until the name AND the mappings are gone."""
    import os
    for path in list(paths):
        for cand in (path, path + ".part"):
            try:
                os.unlink(cand)
            except (FileNotFoundError, IsADirectoryError):
                pass
    del paths[:]


def vb_factorize_sharded(mat, ranks=2, nrun=1, verbose=0, initializer="random", Itmax=10000,
                         hyper_update=(True, True, True, True), gamma_a=1, gamma_b=1, Tol=1e-5,
                         hyper_update_n0=10, hyper_update_dn=1, fudge=None, unif_stop=True, seed=0,
                         device=None, group=None, engine_factory=None, geometry_classes=1, timings=None, concurrent=4):
    """``vb_factorize`` with the (run, rank) units sharded over the ranks of a process group.

    Call it from every process (``torch.distributed`` initialised, one process per GPU).  Every process returns the
    same ``VBResult``.  ``seed`` must be given (not None) so all processes draw the same initial states.  Without a
    process group it runs everything locally.

    What the node's host does ONCE, not once per process (the reference ships the whole bundle to every MPI slave and
    gathers the per-run lists through Rmpi, reference R/bayesian.R:252-263):

    * ``mat`` may be ``None`` on all processes of a node but one: the holder's ``CountMatrix`` is the only ingestion;
      the others run their units on a *shell* (metadata, no entries);
    * the tiled layouts of the sweep (one pair per rank class) are cut by the process(es) that hold X -- side by side
      when several do -- written into /dev/shm and imported by everybody else (``ccfindr_amd.node``);
    * a unit's factor matrices go from its engine straight into a shared result segment (``vbnmf_engine_get_state``
      writes into the mapping); what travels through the process group is one small fp64 tensor -- evidence,
      hyper-parameters, step counts, flags -- so every process assembles the same result from views into that segment.

    Processes on other nodes (no shared /dev/shm) need their own holder of X; the factor matrices of units run on
    another node arrive by tensor broadcast from their owner.  ``timings`` (a dict) receives this process's split of
    the call: ``layout_s`` (cut / export / wait / import), ``units_s``, ``gather_s``.  ``concurrent`` > 1 keeps that many of
    this process's units in flight on its GPU (one engine, HIP stream and host thread each, as ``vb_factorize``): the
    host side of one unit -- initial state, its upload, the result's download -- then runs beside another unit's stepping
    (C4 rehearsal on one GPU, one process: 1.08 s one at a time, 0.73 s with two, 0.60-0.66 s with four); results do not depend
    on it.  Default 4.
    """
    import os
    import time
    import uuid
    import torch
    import torch.distributed as dist
    from . import node as shm
    from .engine import CountMatrix, device_warmup, geometry_rank_for, rank_classes, sweep_workgroups
    t_begin = time.perf_counter()
    world, me = 1, 0
    if dist.is_available() and dist.is_initialized():
        world, me = dist.get_world_size(group), dist.get_rank(group)
    if seed is None:
        raise ValueError("a sharded run needs an explicit seed")
    if device is None:
        device = me
    native = engine_factory is None
    X = None
    if mat is not None:
        X = mat if isinstance(mat, CountMatrix) or not native else CountMatrix(mat)
    # ---- control plane, round 1 (tiny host objects): who holds X, on which node, and what the guards found
    mine = {"node": shm.node_key(), "holds": X is not None, "meta": None, "empty": (0, 0), "n_wg": 0,
            "token": f"{os.getpid()}_{uuid.uuid4().hex[:10]}", "shm_free": shm.free_bytes()}
    if X is not None and native:
        mine["meta"] = [float(v) for v in X.meta()]
        mine["empty"] = (0, 0) if X.is_shell else tuple(X.empty_counts())            # reference R/bayesian.R:244-247
    elif X is not None:
        mine["meta"] = [float(v) for v in (X.shape if hasattr(X, "shape") else np.asarray(X).shape)]
    warm = None
    if native:
        import threading
        mine["n_wg"] = sweep_workgroups(device)
        # the device's first use by this process (context, first allocation, code object: ~0.15 s) on a second host thread:
        # it runs beside the holder's cut, resp. beside this process's wait for the layouts
        warm = threading.Thread(target=device_warmup, args=(device,))
        warm.start()
    t_first = time.perf_counter()
    peers = [mine]
    if world > 1:
        peers = [None] * world
        dist.all_gather_object(peers, mine, group=group)
    holders = [p for p in range(world) if peers[p]["holds"]]
    if not holders:
        raise ValueError("no process holds the count matrix (mat is None everywhere)")
    for p in holders:                                    # the guards of vb_factorize, raised by EVERY process
        if peers[p]["empty"][0] > 0:
            raise ValueError("Input matrix contains empty rows")
        if peers[p]["empty"][1] > 0:
            raise ValueError("Input matrix contains empty columns")
    # Room in each node's memory file system for what is about to be put there (a container may give /dev/shm 64 MB: writing
    # past a full tmpfs is a SIGBUS, not an error code).  Estimate: per geometry two entry streams of 4 B (12 B for
    # non-integer X) per stored entry + 15 %, and the units' four factor matrices.  A node with too little room is taken
    # apart: each of its processes then works alone (its own layouts, results by tensor broadcast), which needs X on all of
    # them.  Every process evaluates every node from the same gathered numbers, so all take the same decisions.
    any_meta = next(peers[p]["meta"] for p in range(world) if peers[p]["holds"])
    geoms_n = max(1, len(set(int(r) for r in np.atleast_1d(ranks))) if not geometry_classes else int(geometry_classes))
    entry_b = 4 if (len(any_meta) > 3 and any_meta[3]) else 12
    need = sum(16.0 * (any_meta[0] * r + r * any_meta[1]) for r in np.atleast_1d(ranks)) * nrun
    if native:
        need += geoms_n * 2 * (float(any_meta[2]) * entry_b * 1.15 + 64e6)
    for key in sorted({q["node"] for q in peers}):
        members = [p for p in range(world) if peers[p]["node"] == key]
        if len(members) > 1 and min(peers[p]["shm_free"] for p in members) < 1.25 * need:
            lacking = [p for p in members if not peers[p]["holds"]]
            if lacking:
                raise RuntimeError(f"processes {lacking} hold no copy of X and the memory file system of their node has room for "
                                   f"{min(peers[p]['shm_free'] for p in members) / 1e9:.2f} GB of the {1.25 * need / 1e9:.2f} GB its "
                                   "processes would share; give every process the matrix or point VBNMF_SHM_DIR at a larger one")
            for p in members:
                peers[p]["node"] = f"{key}#alone{p}"
    mine = peers[me]
    my_node = [p for p in range(world) if peers[p]["node"] == mine["node"]]
    node_holders = [p for p in my_node if peers[p]["holds"]]
    if not node_holders:
        raise ValueError(f"process {me}: no process of this node holds the count matrix; every node needs one holder")
    meta = peers[node_holders[0]]["meta"]
    if X is None and native:
        X = CountMatrix.shell(meta)
    n, m = int(meta[0]), int(meta[1])
    bundle = make_bundle(X if native else mat, ranks, nrun, verbose, initializer, Itmax, hyper_update, gamma_a, gamma_b, Tol,
                         hyper_update_n0, hyper_update_dn, fudge, unif_stop, seed, device, engine_factory,
                         check_empty=not native) \
        if (native or mat is not None) else None                   # (native: the guards ran above, on the holders, for everybody)
    if bundle is None:
        raise ValueError("an injected engine_factory needs the matrix on every process")
    tasks, costs = sweep_tasks(bundle["ranks"], nrun)
    schedule = lpt_schedule(costs, world)
    owner_of = {t: p for p, ts in enumerate(schedule) for t in ts}
    bundle["engines"] = {} if nrun > 1 else None        # this process's restarts of a rank share the engine
    if timings is not None:
        bundle["unit_times"] = []
    plan_geometry(bundle, geometry_classes)

    # ---- the sweep's layouts: cut once per node, by its holders side by side, shared through /dev/shm.
    # No collective carries them: a segment's name follows from its builder's token (round 1), the builder renames the
    # finished segment into place and the peers poll for the name -- so a peer imports the cell side while the builder is
    # still cutting the gene side, and the builder writes one side out (a second host thread) while it cuts the next.
    detail = {} if timings is not None else None
    tick = time.perf_counter
    cleanup = []                                         # files this process created (unlinked behind the next barrier)
    layout_error = None
    if native:
        try:
            import threading
            geoms = sorted({geometry_rank_for(r, bundle["classes"]) or int(r) for r in bundle["ranks"]})
            pieces = [(g, side) for g in geoms for side in (1, 0)]          # cell side first: it needs no row-major copy of X
            builder_of = {pc: node_holders[q % len(node_holders)] for q, pc in enumerate(pieces)}
            n_wg = mine["n_wg"]
            sharing = world > 1 and len(my_node) > 1
            seg_name = lambda b, pc: f"vbnmf_{peers[b]['token']}_g{pc[0]}_s{pc[1]}"
            failed_name = lambda b: f"vbnmf_{peers[b]['token']}_failed"
            wait_s = float(os.environ.get("VBNMF_WAIT_TIMEOUT_S", "300") or 300)
            seg_path = lambda b, pc: os.path.join(shm.shm_dir(), seg_name(b, pc))
            if sharing and any(b == me for b in builder_of.values()):
                loaders, errors = [], []

                def load(pc):                                # this process's own device copy, beside the cut of the next piece
                    try:
                        t0 = tick()
                        X.preload_layout(pc[1], pc[0], n_wg, device)
                        if detail is not None:
                            detail[f"preload_side{pc[1]}_s"] = tick() - t0
                    except BaseException as exc:             # noqa: BLE001
                        errors.append(exc)

                # The peers of this node wait for these cuts and their cores idle: the builder takes its share of the node's cores
                # for the duration (the library's default stops at 32 threads per process, a cap meant for ranks that all work at
                # once).  Layouts and cell order do not depend on the thread count.
                from .engine import host_threads, set_host_threads
                builders_here = max(1, len(set(b for b in builder_of.values() if b in my_node)))
                cores = shm.usable_cores()                   # (affinity mask AND cgroup quota)
                cores = int(os.environ.get("VBNMF_TEST_NODE_CORES") or 0) or cores        # test hook: pretend the node has this many
                want = min(128, cores // builders_here)
                lifted = want > host_threads() and not os.environ.get("VBNMF_HOST_THREADS")
                if lifted:
                    set_host_threads(want)
                if detail is not None:
                    detail["cut_threads"] = host_threads()
                try:
                    X.prepare_async()                        # cell order, then the row-major copy, beside the cut of the cell side
                    for pc in pieces:
                        if builder_of[pc] != me:
                            continue
                        t0 = tick()
                        cleanup.append(seg_path(me, pc))                         # (registered first: a failed cut leaves a .part)
                        X.share_layout(pc[1], pc[0], n_wg, seg_path(me, pc))         # cut INTO the shared file; the name appears when complete
                        if detail is not None:
                            detail[f"cut_side{pc[1]}_s"] = tick() - t0
                        th = threading.Thread(target=load, args=(pc,))
                        th.start()
                        loaders.append(th)
                    for th in loaders:
                        th.join()
                    if errors:
                        raise errors[0]
                except BaseException as exc:
                    with open(os.path.join(shm.shm_dir(), failed_name(me)), "w") as fh:   # the peers stop polling and raise
                        fh.write(f"{type(exc).__name__}: {exc}")
                    cleanup.append(os.path.join(shm.shm_dir(), failed_name(me)))
                    raise
                finally:
                    if lifted:
                        set_host_threads(0)
            if sharing:
                t0 = tick()
                for pc in pieces:
                    b = builder_of[pc]
                    if b == me:
                        continue
                    if peers[b]["n_wg"] != n_wg:
                        raise RuntimeError(f"process {b} cuts layouts for {peers[b]['n_wg']} workgroups, this device wants {n_wg}")
                    t1 = tick()
                    shm.wait_for(seg_name(b, pc), wait_s, failed_name(b))
                    t2 = tick()
                    X.attach_layout(seg_path(b, pc))                 # mapped, not copied
                    t3 = tick()
                    X.preload_layout(pc[1], pc[0], n_wg, device)     # upload it now, beside the wait for the next piece
                    if detail is not None:
                        detail[f"wait_side{pc[1]}_s"] = t2 - t1
                        detail[f"attach_side{pc[1]}_s"] = t3 - t2
                        detail[f"preload_side{pc[1]}_s"] = tick() - t3
                if detail is not None:
                    detail["wait_and_attach_s"] = tick() - t0
            if not sharing and int(concurrent) > 1 and not X.is_shell:
                # nobody to share with, but several units in flight: cut (and upload) the sweep's layouts once, here, instead of
                # letting the first units' threads cut the same pair side by side
                # (the two sides of a geometry side by side, as engine creation cuts them: each cut is memory-bound before it uses
                # every host thread, and a side's upload runs beside the other side's cut)
                X.prepare_async()
                errs = []

                def cut(pc):
                    try:
                        X.preload_layout(pc[1], pc[0], n_wg, device)
                    except BaseException as exc:             # noqa: BLE001
                        errs.append(exc)
                ths = [threading.Thread(target=cut, args=(pc,)) for pc in pieces]
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
                if errs:
                    raise errs[0]
        except BaseException as exc:                      # noqa: BLE001 -- carried through the process group below
            layout_error = exc
    if warm is not None:
        warm.join()
    # A layout phase that failed on ONE process (a builder's cut, a peer's wait or n_wg check) must not leave the others in
    # the collectives below until the backend's timeout: the failure travels as a flag every process reaches, the files
    # this process put into the node's memory file system go (finished segments, a half-written .part, the failure note
    # the node's pollers have read by now), and every process raises.
    failed_here = None if layout_error is None else f"{type(layout_error).__name__}: {layout_error}"
    if _any_failed(failed_here is not None, world, group, device):
        notes = [failed_here]
        if world > 1:
            notes = [None] * world
            dist.all_gather_object(notes, failed_here, group=group)
        _remove_files(cleanup)
        if layout_error is not None:
            raise layout_error
        who = next(p for p in range(world) if notes[p] is not None)
        raise ShardedRunError(f"the layout phase failed on process {who}: {notes[who]}")
    t_layout = time.perf_counter()

    # ---- the result segment of this node: [ew | eh | sdw | sdh] of every unit, written by the unit's owner
    offsets, total = _unit_offsets(tasks, n, m)
    rseg = None
    if world > 1:
        # EVERY process takes part in these collectives, whatever its node looks like (a node with one process, or one taken
        # apart for lack of room, shares nothing but must not leave the others waiting: nodes {0,1} + {2} used to hang here)
        leader = my_node[0]
        shares = len(my_node) > 1
        seg_error = None
        if shares and me == leader:
            try:
                rseg = shm.Segment.create(shm.fresh_name("results"), total)
            except BaseException as exc:                 # noqa: BLE001
                seg_error = f"{type(exc).__name__}: {exc}"
        box = [None] * world
        dist.all_gather_object(box, (rseg.name if (shares and me == leader and rseg is not None) else None, seg_error), group=group)
        if shares and me != leader and box[leader][0] is not None:
            try:
                rseg = shm.Segment.open(box[leader][0])
            except BaseException as exc:                 # noqa: BLE001
                seg_error = f"{type(exc).__name__}: {exc}"
        bad = seg_error is not None or any(b[1] is not None for b in box)
        # (the flag's all-reduce is also the barrier: behind it every process of every node has imported the layouts and
        # mapped the results)
        bad = _any_failed(bad, world, group, device)
        if rseg is not None:
            rseg.unlink()                                # mapped everywhere: the name can go, the memory lives with the mappings
        _remove_files(cleanup)                           # every peer holds its mapping: the names can go
        if bad:
            first = next((f"process {p}: {b[1]}" for p, b in enumerate(box) if b[1] is not None), seg_error)
            raise ShardedRunError(f"the result segment could not be set up ({first})")
        if rseg is not None:
            bundle["state_out"] = lambda irun, r: _unit_views(rseg, offsets[tasks.index((irun, int(r)))], n, m, int(r))

    # A unit that raises (hyper-parameter Newton failure, a VBNMFError, rank > min(nrow, ncol) ...) must not keep this
    # process from the collectives below: the others would wait in them for ever.  The error travels as a flag, every
    # process reaches the collectives, and then every process raises the first error (by unit order).
    rmax = max([r for _, r in tasks] + [1])
    rec = torch.zeros((len(tasks), _REC_UNIF0 + rmax), dtype=torch.float64)
    local, failure = {}, None
    bundle["concurrent"] = max(1, int(concurrent))      # (restarts of a rank, nrun > 1, keep one engine per thread and rank)

    def run_unit(t):
        """One unit -> its record row filled; returns the failure record or None."""
        try:
            out = vb_run_rank(tasks[t][0], tasks[t][1], bundle)
        except Exception as exc:                                     # noqa: BLE001 -- re-raised after the exchange
            rec[t, _REC_FAIL] = 1.0
            return (t, me, type(exc).__name__, str(exc))
        local[t] = out
        row = rec[t]
        row[_REC_DONE] = 1.0; row[_REC_LK0] = out["lk0"]; row[_REC_NSTEPS] = out["nsteps"]; row[_REC_OWNER] = me
        for q, key in enumerate(("aw", "bw", "ah", "bh")):
            row[_REC_AW + q] = out["hyper"][key]
        for c in out["unif"]:
            row[_REC_UNIF0 + c - 1] = 1.0
        return None

    try:
        if bundle["concurrent"] > 1 and len(schedule[me]) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=bundle["concurrent"]) as pool:
                fails = [f for f in pool.map(run_unit, schedule[me]) if f is not None]
            failure = min(fails) if fails else None
        else:
            for t in schedule[me]:
                failure = run_unit(t)
                if failure is not None:
                    break
    finally:
        _close_engines(bundle)
    t_units = time.perf_counter()

    records = {}
    if world > 1:
        # one small tensor through the process group: every row is written by exactly one process, so the sum IS the gather
        backend = dist.get_backend(group)
        buf = rec.to(torch.device("cuda", device)) if backend == "nccl" else rec
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        rec = buf.cpu()
        if float(rec[:, _REC_FAIL].sum()) > 0:
            fails = [None] * world
            dist.all_gather_object(fails, failure, group=group)              # (rare path: the messages are host strings)
            t, who, kind, msg = min(f for f in fails if f is not None)
            irun, r = tasks[t]
            raise ShardedRunError(f"unit (run {irun}, rank {r}) failed on process {who}: {kind}: {msg}")
        for t, (irun, r) in enumerate(tasks):
            row = rec[t]
            if row[_REC_DONE] == 0:
                continue
            if t in local:
                mats = local[t]
            elif rseg is not None and owner_of[t] in my_node:
                v = _unit_views(rseg, offsets[t], n, m, int(r))
                mats = {"ew": v["ew"], "eh": v["eh"], "sdw": v["dw"], "sdh": v["dh"]}
            else:
                mats = None                                          # another node's unit: fetched below
            records[(irun, r)] = {"rank": r, "lk0": float(row[_REC_LK0]), "nsteps": int(row[_REC_NSTEPS]),
                                  "hyper": {k: float(row[_REC_AW + q]) for q, k in enumerate(("aw", "bw", "ah", "bh"))},
                                  "unif": [c + 1 for c in range(int(r)) if row[_REC_UNIF0 + c] != 0], "_mats": mats}
        # units of other nodes: their four matrices by tensor broadcast from the owner (no pickling)
        remote = [t for t, (irun, r) in enumerate(tasks) if (irun, r) in records and
                  any(peers[p]["node"] != peers[owner_of[t]]["node"] for p in range(world))]
        for t in remote:
            irun, r = tasks[t]
            src = owner_of[t]
            dev = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")
            flat = torch.empty(2 * (n * r + r * m), dtype=torch.float64, device=dev)
            if me == src:
                o = local[t]
                flat.copy_(torch.from_numpy(np.concatenate([np.asarray(o[k]).ravel(order="F") for k in ("ew", "eh", "sdw", "sdh")])))
            dist.broadcast(flat, src=src if group is None else dist.get_global_rank(group, src), group=group)
            if records[(irun, r)]["_mats"] is None:
                a = flat.cpu().numpy()
                nr, rm = n * r, r * m
                records[(irun, r)]["_mats"] = {"ew": a[:nr].reshape((n, r), order="F"), "eh": a[nr:nr + rm].reshape((r, m), order="F"),
                                               "sdw": a[nr + rm:2 * nr + rm].reshape((n, r), order="F"),
                                               "sdh": a[2 * nr + rm:].reshape((r, m), order="F")}
        for key, recd in records.items():
            recd.update(recd.pop("_mats"))
    else:
        if failure is not None:
            t, who, kind, msg = failure
            irun, r = tasks[t]
            raise ShardedRunError(f"unit (run {irun}, rank {r}) failed on process {who}: {kind}: {msg}")
        records = {tasks[t]: out for t, out in local.items()}
    vb = []
    for irun in range(1, nrun + 1):
        per_rank = {r: records[(irun, r)] for r in bundle["ranks"] if (irun, r) in records}
        vb.append(assemble_run(per_rank, bundle["ranks"], unif_stop))
    res = select_best(vb, bundle["ranks"])
    if timings is not None:
        t_end = time.perf_counter()
        timings.update({"layout_s": t_layout - t_begin, "units_s": t_units - t_layout, "gather_s": t_end - t_units,
                        "total_s": t_end - t_begin, "units": len(schedule[me]), "node_processes": len(my_node),
                        "node_holders": len(node_holders), "is_shell": bool(native and X.is_shell),
                        "layout_detail": (detail if native else None), "first_call_s": t_first - t_begin,
                        "unit_detail": bundle.get("unit_times")})
    return res


# ---------------------------------------------------------------------------------------
# cell-partitioned single factorisation
# ---------------------------------------------------------------------------------------
def cell_partition(m, world):
    """Contiguous, near-equal column blocks: [(begin, end)] * world."""
    return [(m * k // world, m * (k + 1) // world) for k in range(world)]


class CellPartitionedEngine:
    """One factorisation over all GPUs of a process group, cells partitioned.

    Same surface as ``VBEngine`` (``set_state``, ``step``, ``run``, ``get_state``); ``lh``/``eh`` arguments
    and results are FULL r x m matrices, each process uses / returns its own column block
    (``get_state`` all-gathers the blocks).

    The per-step all-reduce is the LIBRARY's (``native=True``, the default on an RCCL process group or without one):
    rank 0 draws an RCCL id, ``torch.distributed`` broadcasts it once (control plane), every process builds a
    ``vbnmf_comm`` and attaches it to its partition engine; ``step`` is then step_local / vbnmf_engine_allreduce /
    step_finish and ``run`` the device-driven loop of vbnmf_engine_run, with no Python and no torch between steps.
    ``native=False`` keeps the exchange in ``torch.distributed`` on the engine's stream (gloo rehearsals of the
    multi-process path on one GPU -- RCCL refuses two ranks on a device -- and the CPU tests, which inject a numpy
    engine through ``engine``); there is no device-driven loop on that path.
    """

    def __init__(self, X, rank, device=0, group=None, engine=None, native=None, block=None):
        """block = (col_begin, col_end, m_global): X holds ONLY this process's column block (m_local = col_end - col_begin
        columns) -- a process of a node-shared run ingests its own cells and nothing else; the blocks must be
        cell_partition(m_global, world).  Default: X is the whole matrix and the engine cuts its block out of it."""
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.me = dist.get_rank(group) if inited else 0
        n, m = X.shape
        if block is not None:
            cb, ce, m = int(block[0]), int(block[1]), int(block[2])
            if (cb, ce) != cell_partition(m, self.world)[self.me] or X.shape[1] != ce - cb:
                raise ValueError(f"block {block} is not this process's share of {m} cells over {self.world} processes")
        self.n, self.m_global, self.rank = n, m, int(rank)
        self.cols = cell_partition(m, self.world)[self.me]
        self.m = self.cols[1] - self.cols[0]
        injected = engine is not None
        if engine is None:
            engine = (VBEngine(X, rank, device=device, cols=(0, self.m), m_global=m) if block is not None
                      else VBEngine(X, rank, device=device, cols=self.cols, m_global=m))
        self.engine = engine
        if native is None:
            # one process = no exchange: no communicator is built (a box without librccl can still run it) and the
            # engine, unpartitioned, drives its own loop
            # (VBNMF_RCCL_LIB names a stand-in for librccl that accepts several ranks on one device -- tests/fake_rccl --, so a
            # gloo rehearsal on one GPU can still run the library's own collective and its device-driven loop)
            import os
            native = (not injected) and self.world > 1 and (dist.get_backend(group) == "nccl" or bool(os.environ.get("VBNMF_RCCL_LIB")))
        self.native = bool(native)
        self.comm = None
        if self.native:
            from .engine import Communicator
            box = [Communicator.unique_id() if self.me == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(box, src=0, group=group)          # control plane, once per engine
            self.comm = Communicator.rccl(box[0], self.world, self.me, device)
            engine.attach_comm(self.comm)
            self._red = None
        else:
            self._red = engine.reduce_tensor()
        self._stream_ctx = getattr(engine, "stream_context", None)

    def _allreduce(self):
        if self.native:
            self.engine.allreduce()
            return
        if self.world == 1:
            return
        red = self._red
        staged = getattr(red, "is_cuda", False) and self._dist.get_backend(self.group) != "nccl"
        if staged:
            # a backend that cannot reduce device memory (gloo: rehearsals of the multi-process path on one GPU):
            # the buffer goes through the host, in order on the engine's stream
            with self._stream_ctx():
                host = red.cpu()
            self._dist.all_reduce(host, op=self._dist.ReduceOp.SUM, group=self.group)
            with self._stream_ctx():
                red.copy_(host)
        elif self._stream_ctx is not None:
            with self._stream_ctx():
                self._dist.all_reduce(red, op=self._dist.ReduceOp.SUM, group=self.group)
        else:
            self._dist.all_reduce(red, op=self._dist.ReduceOp.SUM, group=self.group)

    def set_state(self, lw, lh, eh):
        cb, ce = self.cols
        self.engine.set_state(lw, np.asarray(lh)[:, cb:ce], np.asarray(eh)[:, cb:ce])
        if self.world > 1 or self.native and self.m != self.m_global:   # an unpartitioned engine finishes set_state by itself
            self._allreduce()
            self.engine.state_finish()

    def step(self, hyper, fudge=EPS):
        self.engine.step_local(hyper, fudge)
        self._allreduce()
        return self.engine.step_finish()

    def run(self, hyper, **kw):
        """The device-driven loop of ``VBEngine.run`` across the partitions (native communicator only): every
        process calls it with the same arguments and gets the same result."""
        if not self.native and not (self.world == 1 and hasattr(self.engine, "run") and self.m == self.m_global):
            raise RuntimeError("the device-driven loop of a partitioned run needs the native (RCCL) communicator")
        return self.engine.run(hyper, **kw)

    def _gather_cells(self, a):
        """r x m_local blocks -> the full r x m matrix on every process (one tensor all_gather, blocks padded to the
        widest partition; no pickling)."""
        torch, dist = self._torch, self._dist
        counts = [e - b for b, e in cell_partition(self.m_global, self.world)]
        on_gpu = dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", self.engine.device) if on_gpu else torch.device("cpu")
        mine = torch.zeros((max(counts), a.shape[0]), dtype=torch.float64, device=dev)
        mine[:a.shape[1]] = torch.from_numpy(np.ascontiguousarray(a.T)).to(dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        return np.concatenate([p[:c].cpu().numpy().T for p, c in zip(parts, counts)], axis=1)

    def get_state(self, names=("lw", "lh", "ew", "eh", "dw", "dh")):
        local = self.engine.get_state(names)
        if self.world == 1:
            return local
        out = {k: v for k, v in local.items() if k in ("lw", "ew", "dw")}
        for k in ("lh", "eh", "dh"):
            if k in local:
                out[k] = self._gather_cells(local[k])
        return out

    def close(self):
        self.engine.close()
        if self.comm is not None:
            self.comm.close()


def _hip_stream_context(engine):
    """Context manager making the engine's HIP stream torch's current stream, so an RCCL
    collective issued inside is ordered after step_local's kernels and before step_finish's."""
    import torch
    ext = torch.cuda.ExternalStream(engine.stream(), device=torch.device("cuda", engine.device))
    return lambda: torch.cuda.stream(ext)


# VBEngine gains the context lazily (torch is only needed for multi-GPU runs)
def _vbengine_stream_context(self):
    ctx = getattr(self, "_stream_ctx_factory", None)
    if ctx is None:
        ctx = self._stream_ctx_factory = _hip_stream_context(self)
    return ctx()


VBEngine.stream_context = _vbengine_stream_context
