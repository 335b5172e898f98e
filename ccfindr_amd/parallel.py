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


def vb_factorize_sharded(mat, ranks=2, nrun=1, verbose=0, initializer="random", Itmax=10000,
                         hyper_update=(True, True, True, True), gamma_a=1, gamma_b=1, Tol=1e-5,
                         hyper_update_n0=10, hyper_update_dn=1, fudge=None, unif_stop=True, seed=0,
                         device=None, group=None, engine_factory=None, geometry_classes=1):
    """``vb_factorize`` with the (run, rank) units sharded over the ranks of a process group.

    Call it from every process (``torch.distributed`` initialised, one process per GPU).
    Every process returns the same ``VBResult``.  ``seed`` must be given (not None) so all
    processes draw the same initial states.  Without a process group it runs everything locally.
    """
    import torch.distributed as dist
    world, me = 1, 0
    if dist.is_available() and dist.is_initialized():
        world, me = dist.get_world_size(group), dist.get_rank(group)
    if seed is None:
        raise ValueError("a sharded run needs an explicit seed")
    if device is None:
        device = me
    bundle = make_bundle(mat, ranks, nrun, verbose, initializer, Itmax, hyper_update, gamma_a, gamma_b, Tol,
                         hyper_update_n0, hyper_update_dn, fudge, unif_stop, seed, device, engine_factory)
    tasks, costs = sweep_tasks(bundle["ranks"], nrun)
    mine = lpt_schedule(costs, world)[me]
    bundle["engines"] = {} if nrun > 1 else None        # this process's restarts of a rank share the engine
    # the geometry plan covers ALL ranks of the sweep, not only this process's units: every process cuts the same
    # layouts, so a unit's result does not depend on which process ran it (bit for bit, as vb_factorize's)
    planned = plan_geometry(bundle, geometry_classes)
    # A unit that raises (hyper-parameter Newton failure, a VBNMFError, rank > min(nrow, ncol) ...) must not keep
    # this process from the gather below: the other processes would wait in it for ever.  The error travels as a
    # record, every process reaches the collective, and then every process raises the first error (by unit order).
    local, failure = {}, None
    try:
        for t in mine:
            try:
                local[tasks[t]] = vb_run_rank(tasks[t][0], tasks[t][1], bundle)
            except Exception as exc:                                 # noqa: BLE001 -- re-raised after the gather
                failure = (t, me, type(exc).__name__, str(exc))
                break
    finally:
        _close_engines(bundle)
        if planned:
            bundle["mat"].plan_ranks(())
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, (local, failure), group=group)   # control plane: host objects, once per sweep
        records = {}
        failures = []
        for part, fail in gathered:
            records.update(part)
            if fail is not None:
                failures.append(fail)
    else:
        records, failures = local, ([failure] if failure is not None else [])
    if failures:
        t, who, kind, msg = min(failures)
        irun, r = tasks[t]
        raise ShardedRunError(f"unit (run {irun}, rank {r}) failed on process {who}: {kind}: {msg}")
    vb = []
    for irun in range(1, nrun + 1):
        per_rank = {r: records[(irun, r)] for r in bundle["ranks"] if (irun, r) in records}
        vb.append(assemble_run(per_rank, bundle["ranks"], unif_stop))
    return select_best(vb, bundle["ranks"])


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

    def __init__(self, X, rank, device=0, group=None, engine=None, native=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.group = group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        self.me = dist.get_rank(group) if inited else 0
        n, m = X.shape
        self.n, self.m_global, self.rank = n, m, int(rank)
        self.cols = cell_partition(m, self.world)[self.me]
        self.m = self.cols[1] - self.cols[0]
        injected = engine is not None
        if engine is None:
            engine = VBEngine(X, rank, device=device, cols=self.cols, m_global=m)
        self.engine = engine
        if native is None:
            # one process = no exchange: no communicator is built (a box without librccl can still run it) and the
            # engine, unpartitioned, drives its own loop
            native = (not injected) and self.world > 1 and dist.get_backend(group) == "nccl"
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
