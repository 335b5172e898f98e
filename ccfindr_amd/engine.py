"""Host-side objects over the C ABI: the ingested count matrix and the device-resident engine.

``CountMatrix`` is what ``counts(object)`` is to the reference driver (reference
R/bayesian.R:239): the genes x cells count matrix, dense or ``dgCMatrix``-like sparse.
``VBEngine`` holds what ``vb_iterate`` carries from one ``vbnmf_update`` call to the next
(``wh``; reference R/bayesian.R:334-339) in HBM and runs the step there.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _native as N

EPS = float(np.finfo(np.float64).eps)    # .Machine$double.eps, the default fudge (reference R/bayesian.R:238)


class CountMatrix:
    """X ingested once into the library's canonical sparse copy (zeros dropped)."""

    def __init__(self, X):
        L = N.load()
        self._h = ctypes.c_void_p()
        self._lib = L
        if hasattr(X, "tocsc") and hasattr(X, "indptr"):          # scipy.sparse
            fmt = getattr(X, "format", None)
            if fmt == "csr":
                n, m = X.shape
                p = np.ascontiguousarray(X.indptr, dtype=np.int32)
                j = np.ascontiguousarray(X.indices, dtype=np.int32)
                x = np.ascontiguousarray(X.data, dtype=np.float64)
                N.check(L.vbnmf_matrix_from_csr(n, m, p.ctypes.data_as(N.c_int32_p), j.ctypes.data_as(N.c_int32_p),
                                                N.dptr(x), ctypes.byref(self._h)))
            else:
                X = X.tocsc()
                n, m = X.shape
                p = np.ascontiguousarray(X.indptr, dtype=np.int32)
                i = np.ascontiguousarray(X.indices, dtype=np.int32)
                x = np.ascontiguousarray(X.data, dtype=np.float64)
                N.check(L.vbnmf_matrix_from_csc(n, m, p.ctypes.data_as(N.c_int32_p), i.ctypes.data_as(N.c_int32_p),
                                                N.dptr(x), ctypes.byref(self._h)))
        else:
            A = N.fcol(X)
            if A.ndim != 2:
                raise ValueError("X must be a 2-d matrix (genes x cells)")
            n, m = A.shape
            N.check(L.vbnmf_matrix_from_dense(n, m, N.dptr(A), ctypes.byref(self._h)))
        n_, m_, nnz_ = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        lgx = ctypes.c_double()
        N.check(L.vbnmf_matrix_info(self._h, ctypes.byref(n_), ctypes.byref(m_), ctypes.byref(nnz_), ctypes.byref(lgx)))
        self.shape = (n_.value, m_.value)
        self.nnz = nnz_.value
        self.sum_lgamma_x1 = lgx.value

    @classmethod
    def from_csc(cls, n, m, p, i, x):
        """dgCMatrix slots: @p, @i, @x (reference R/utils.R:34 reads 10x data into one)."""
        self = cls.__new__(cls)
        L = N.load()
        self._lib = L
        self._h = ctypes.c_void_p()
        p = np.ascontiguousarray(p, dtype=np.int32)
        i = np.ascontiguousarray(i, dtype=np.int32)
        x = np.ascontiguousarray(x, dtype=np.float64)
        N.check(L.vbnmf_matrix_from_csc(n, m, p.ctypes.data_as(N.c_int32_p), i.ctypes.data_as(N.c_int32_p),
                                        N.dptr(x), ctypes.byref(self._h)))
        nnz_ = ctypes.c_int64()
        lgx = ctypes.c_double()
        N.check(L.vbnmf_matrix_info(self._h, None, None, ctypes.byref(nnz_), ctypes.byref(lgx)))
        self.shape = (int(n), int(m))
        self.nnz = nnz_.value
        self.sum_lgamma_x1 = lgx.value
        return self

    @classmethod
    def from_mtx(cls, path):
        """A Matrix Market file, parsed natively into compressed columns (reference R/utils.R:34:
        ``as(Matrix::readMM(count), 'dgCMatrix')``)."""
        self = cls.__new__(cls)
        L = N.load()
        self._lib = L
        self._h = ctypes.c_void_p()
        N.check(L.vbnmf_matrix_from_mtx(os.fsencode(path), ctypes.byref(self._h)))
        n_, m_, nnz_ = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        lgx = ctypes.c_double()
        N.check(L.vbnmf_matrix_info(self._h, ctypes.byref(n_), ctypes.byref(m_), ctypes.byref(nnz_), ctypes.byref(lgx)))
        self.shape = (n_.value, m_.value)
        self.nnz = nnz_.value
        self.sum_lgamma_x1 = lgx.value
        return self

    def write_mtx(self, path):
        """``Matrix::writeMM`` of the counts (reference R/utils.R:876)."""
        N.check(self._lib.vbnmf_matrix_write_mtx(self._h, os.fsencode(path)))

    def to_scipy(self):
        """The canonical compressed columns as a scipy ``csc_matrix`` (a copy)."""
        import scipy.sparse as sp
        cp, ri, xv = N.c_int64_p(), N.c_int32_p(), N.c_double_p()
        N.check(self._lib.vbnmf_matrix_csc(self._h, ctypes.byref(cp), ctypes.byref(ri), ctypes.byref(xv)))
        n, m = self.shape
        indptr = np.ctypeslib.as_array(cp, shape=(m + 1,)).copy()
        indices = np.ctypeslib.as_array(ri, shape=(max(self.nnz, 1),))[:self.nnz].copy()
        data = np.ctypeslib.as_array(xv, shape=(max(self.nnz, 1),))[:self.nnz].copy()
        return sp.csc_matrix((data, indices, indptr), shape=(n, m))

    def empty_counts(self):
        """(# all-zero rows, # all-zero columns): the guards of reference R/bayesian.R:244-247."""
        er, ec = ctypes.c_int64(), ctypes.c_int64()
        N.check(self._lib.vbnmf_matrix_empty_counts(self._h, ctypes.byref(er), ctypes.byref(ec)))
        return er.value, ec.value

    # -- one ingestion / one pair of layouts per node (vbnmf_matrix_get_meta / _shell / _export_layout / _import_layout) ----
    def meta(self):
        """The eight numbers a shell of this matrix needs (dimensions, entry count, value-kind flags, constants)."""
        out = np.empty(8)
        N.check(self._lib.vbnmf_matrix_get_meta(self._h, N.dptr(out)))
        return out

    @classmethod
    def shell(cls, meta):
        """A handle with X's metadata and NO entries: engines on it use layouts imported from the process that holds X
        (``import_layout``); anything that needs the entries raises ``VBNMFError`` (status 5)."""
        self = cls.__new__(cls)
        L = N.load()
        self._lib = L
        self._h = ctypes.c_void_p()
        meta = np.ascontiguousarray(meta, dtype=np.float64)
        N.check(L.vbnmf_matrix_shell(N.dptr(meta), ctypes.byref(self._h)))
        self.shape = (int(meta[0]), int(meta[1]))
        self.nnz = int(meta[2])
        self.sum_lgamma_x1 = float(meta[6])
        return self

    @property
    def is_shell(self):
        return bool(self._lib.vbnmf_matrix_is_shell(self._h))

    def prepare_async(self):
        """The same on a background host thread of the library (returns at once)."""
        N.check(self._lib.vbnmf_matrix_prepare_async(self._h))

    def preload_layout(self, side, geometry_rank, n_wg, device=0):
        """Uploads that whole-matrix layout to the device ahead of the first engine (the engines share the resident copy)."""
        N.check(self._lib.vbnmf_matrix_preload_layout(self._h, int(side), int(geometry_rank), int(n_wg), int(device)))

    def prepare(self):
        """Cell order + row-major copy of X ahead of need (thread-safe; a second host thread may run it beside a cut)."""
        N.check(self._lib.vbnmf_matrix_prepare(self._h))

    def layout_blob_size(self, side, geometry_rank, n_wg):
        """Cuts (and caches) the whole-matrix layout of ``side`` in the geometry of ``geometry_rank`` -> blob bytes."""
        nb = ctypes.c_int64()
        N.check(self._lib.vbnmf_matrix_export_layout(self._h, int(side), int(geometry_rank), int(n_wg), None, 0, ctypes.byref(nb)))
        return nb.value

    def export_layout(self, side, geometry_rank, n_wg, buf):
        """Writes that layout's blob into ``buf`` (a writable buffer, e.g. a shared-memory mapping) -> bytes written."""
        view = (ctypes.c_char * len(buf)).from_buffer(buf)
        nb = ctypes.c_int64()
        N.check(self._lib.vbnmf_matrix_export_layout(self._h, int(side), int(geometry_rank), int(n_wg),
                                                     ctypes.cast(view, ctypes.c_void_p), len(buf), ctypes.byref(nb)))
        return nb.value

    def import_layout(self, buf, nbytes=None):
        """Adds the layout blob in ``buf`` (written by ``export_layout`` in another process of the node) to this handle."""
        nbytes = len(buf) if nbytes is None else int(nbytes)
        try:
            view = (ctypes.c_char * len(buf)).from_buffer(buf)
        except TypeError:                                           # a read-only mapping / bytes
            view = (ctypes.c_char * len(buf)).from_buffer_copy(buf)
        N.check(self._lib.vbnmf_matrix_import_layout(self._h, ctypes.cast(view, ctypes.c_void_p), nbytes))

    def share_layout(self, side, geometry_rank, n_wg, path):
        """Cuts that layout with its entry stream written straight into the new file ``path`` (a memory file system, e.g.
        /dev/shm; complete when the name appears) and keeps the mapping as the layout's storage: one copy per node."""
        N.check(self._lib.vbnmf_matrix_share_layout(self._h, int(side), int(geometry_rank), int(n_wg), os.fsencode(path)))

    def attach_layout(self, path):
        """Maps a layout file written by ``share_layout`` in another process of the node and uses it in place."""
        N.check(self._lib.vbnmf_matrix_attach_layout(self._h, os.fsencode(path)))

    def plan_ranks(self, ranks=(), max_classes=1):
        """Rank classes for a sweep over several ranks (vbnmf_matrix_plan_ranks): engines created afterwards share the
        tiled layouts of the smallest class at or above their rank instead of cutting a pair per LDS row size.  An empty
        ``ranks`` clears the plan."""
        arr = np.asarray(list(ranks), dtype=np.int32)
        N.check(self._lib.vbnmf_matrix_plan_ranks(self._h, arr.ctypes.data_as(N.c_int32_p), int(arr.size), int(max_classes)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.vbnmf_matrix_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rank_classes(ranks, max_classes=1):
    """Rank classes of a sweep (vbnmf_plan_classes): padded ranks, ascending; an engine of rank r takes the geometry of
    the smallest class at or above ``padded_rank(r)``."""
    L = N.load()
    arr = np.asarray(sorted({int(r) for r in ranks}), dtype=np.int32)
    out = np.zeros(max(arr.size, 1), dtype=np.int32)
    cnt = ctypes.c_int32()
    N.check(L.vbnmf_plan_classes(arr.ctypes.data_as(N.c_int32_p), int(arr.size), int(max_classes), out.ctypes.data_as(N.c_int32_p),
                                 ctypes.byref(cnt)))
    return [int(v) for v in out[:cnt.value]]


def geometry_rank_for(rank, classes):
    """The class (a padded rank) an engine of ``rank`` uses, 0 when no class covers it (its own geometry)."""
    pr = int(N.load().vbnmf_padded_rank(int(rank)))
    for c in classes or ():
        if c >= pr:
            return int(c)
    return 0


def sweep_workgroups(device=0):
    """Persistent workgroups of the sweep kernels on ``device``: the ``n_wg`` of its whole-matrix layouts."""
    v = ctypes.c_int32()
    N.check(N.load().vbnmf_device_sweep_workgroups(int(device), ctypes.byref(v)))
    return v.value


def host_threads():
    """Host threads the library's ingestion and layout cuts use in this process (vbnmf_host_threads)."""
    return int(N.load().vbnmf_host_threads())


def set_host_threads(n):
    """Lift or lower the library's host thread count for this process (0: back to the default rule -- the cores of the
    affinity mask, at most 32); returns the count in force before (vbnmf_set_host_threads)."""
    return int(N.load().vbnmf_set_host_threads(int(n)))


def device_warmup(device=0):
    """First use of the device by this process, ahead of need (vbnmf_device_warmup)."""
    N.check(N.load().vbnmf_device_warmup(int(device)))


class _CudaBuffer:
    """Exposes a raw device pointer through __cuda_array_interface__ (for torch.as_tensor)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class VBEngine:
    """Device-resident state of one factorisation (one rank, one column block of X)."""

    def __init__(self, X: CountMatrix, rank: int, device: int = 0, cols=None, m_global=None, geometry_rank: int = 0, grid=None,
                 pad_rank=None):
        """``geometry_rank`` (>= rank): the rank whose LDS row size the tiled layouts are cut for -- the ranks of a sweep
        share one pair of layouts (``rank_classes``); 0: the matrix's plan (``CountMatrix.plan_ranks``) or the rank's own.
        ``grid`` = (sweep workgroups, update blocks): an engine meant for a batch of B (``run_batch``) wants 256 / B of each
        (``vbnmf_set_engine_grid``); None: one per CU.  ``pad_rank``: the engine's factors are stored that many columns wide (a
        padded rank >= this rank's own, ``vbnmf_set_engine_padding``): engines of different ranks made with one ``pad_rank`` may
        share a batch; None: the rank's own width."""
        L = N.load()
        self._lib = L
        self._h = ctypes.c_void_p()
        self.X = X
        cb, ce = (0, X.shape[1]) if cols is None else cols
        mg = X.shape[1] if m_global is None else m_global
        try:
            if grid is not None:
                N.check(L.vbnmf_set_engine_grid(int(grid[0]), int(grid[1])))
            if pad_rank:
                N.check(L.vbnmf_set_engine_padding(int(pad_rank)))
            N.check(L.vbnmf_engine_create_geom(X._h, int(cb), int(ce), int(mg), int(rank), int(geometry_rank), int(device),
                                               ctypes.byref(self._h)))
        finally:
            if grid is not None:
                L.vbnmf_set_engine_grid(0, 0)
            if pad_rank:
                L.vbnmf_set_engine_padding(0)
        n, m, r = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
        N.check(L.vbnmf_engine_dims(self._h, ctypes.byref(n), ctypes.byref(m), ctypes.byref(r)))
        self.n, self.m, self.rank = n.value, m.value, r.value
        self.device = int(device)

    # -- state ---------------------------------------------------------------------------
    def set_state(self, lw, lh, eh, finish=True):
        lw, lh, eh = N.fcol(lw), N.fcol(lh), N.fcol(eh)
        if lw.shape != (self.n, self.rank) or lh.shape != (self.rank, self.m) or eh.shape != (self.rank, self.m):
            raise ValueError(f"state shapes must be lw {(self.n, self.rank)}, lh/eh {(self.rank, self.m)}")
        N.check(self._lib.vbnmf_engine_set_state(self._h, N.dptr(lw), N.dptr(lh), N.dptr(eh)))

    def state_finish(self):
        N.check(self._lib.vbnmf_engine_state_finish(self._h))

    supports_state_out = True

    def get_state(self, names=("lw", "lh", "ew", "eh", "dw", "dh"), out=None):
        """The named members of ``wh`` as column-major arrays.  ``out``: a mapping name -> preallocated column-major fp64
        array of the member's shape (e.g. views into shared memory): the library writes straight into it."""
        res = {}
        for k in ("lw", "ew", "dw", "lh", "eh", "dh"):
            shape = (self.n, self.rank) if k in ("lw", "ew", "dw") else (self.rank, self.m)
            if k not in names:
                res[k] = None
            elif out is not None and k in out:
                a = out[k]
                if a.shape != shape or a.dtype != np.float64 or not a.flags.f_contiguous or not a.flags.writeable:
                    raise ValueError(f"out['{k}'] must be a writable column-major float64 array of shape {shape}")
                res[k] = a
            else:
                res[k] = np.empty(shape, order="F")
        N.check(self._lib.vbnmf_engine_get_state(self._h, N.dptr(res["lw"]), N.dptr(res["lh"]), N.dptr(res["ew"]),
                                                 N.dptr(res["eh"]), N.dptr(res["dw"]), N.dptr(res["dh"])))
        return {k: v for k, v in res.items() if v is not None}

    # -- stepping ------------------------------------------------------------------------
    def step(self, hyper, fudge=EPS):
        """One vbnmf_update on the resident state -> (lkh, stats) with
        stats = (mean log lw, mean log lh, mean ew, mean eh)."""
        lkh = ctypes.c_double()
        st = (ctypes.c_double * 4)()
        N.check(self._lib.vbnmf_engine_step(self._h, hyper["aw"], hyper["bw"], hyper["ah"], hyper["bh"], float(fudge),
                                            ctypes.byref(lkh), st))
        return lkh.value, tuple(st)

    # -- maximum-likelihood NMF on the same engine (reference R/factorize.R:2-27, :40-49) ----
    def ml_set_state(self, w, h):
        w, h = N.fcol(w), N.fcol(h)
        if w.shape != (self.n, self.rank) or h.shape != (self.rank, self.m):
            raise ValueError(f"state shapes must be w {(self.n, self.rank)}, h {(self.rank, self.m)}")
        N.check(self._lib.vbnmf_engine_ml_set_state(self._h, N.dptr(w), N.dptr(h)))

    def ml_step(self, prior=False, gamma_a=1.0, gamma_b=1.0):
        """One nmf_updateR step on the resident (w, h) -> likelihood of the updated pair."""
        lk = ctypes.c_double()
        N.check(self._lib.vbnmf_engine_ml_step(self._h, int(bool(prior)), float(gamma_a), float(gamma_b), ctypes.byref(lk)))
        return lk.value

    def ml_run(self, Itmax=10000, Tol=1e-5, prior=False, gamma_a=1.0, gamma_b=1.0, history=False):
        """factorize()'s likelihood-criterion loop on the resident pair, driven by the device ->
        ``{"it", "lk", "reason"[, "history"]}`` (reason 2 converged, 4 Itmax)."""
        it, reason, lk = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
        hist = np.empty(int(Itmax)) if history else None
        N.check(self._lib.vbnmf_engine_ml_run(self._h, int(bool(prior)), float(gamma_a), float(gamma_b), int(Itmax), float(Tol),
                                              ctypes.byref(it), ctypes.byref(lk), ctypes.byref(reason), N.dptr(hist),
                                              int(Itmax) if history else 0))
        out = {"it": it.value, "lk": lk.value, "reason": reason.value}
        if history:
            out["history"] = hist[:it.value].copy()
        return out

    def ml_likelihood(self):
        """likelihood(mat, w, h) of the pair the engine holds now."""
        lk = ctypes.c_double()
        N.check(self._lib.vbnmf_engine_ml_likelihood(self._h, ctypes.byref(lk)))
        return lk.value

    def ml_get_state(self, names=("ew", "eh")):
        w = np.empty((self.n, self.rank), order="F") if "ew" in names else None
        h = np.empty((self.rank, self.m), order="F") if "eh" in names else None
        N.check(self._lib.vbnmf_engine_ml_get_state(self._h, N.dptr(w), N.dptr(h)))
        return {k: v for k, v in (("ew", w), ("eh", h)) if v is not None}

    def cluster_ids(self):
        """1-based arg-max component of every cell (``which.max(h[, j])``) of the ML / VB state the engine holds."""
        ids = np.empty(self.m, dtype=np.int32)
        N.check(self._lib.vbnmf_engine_cluster_ids(self._h, ids.ctypes.data_as(N.c_int32_p)))
        return ids

    def cluster_changes(self, want_ids=False):
        """``sum(cnn != cnn0)`` between the labels of the state held now and those of the previous call (reference
        R/factorize.R:198-208), counted on the device; ``None`` on the first call after a state was loaded.
        Returns ``(changed, ids or None)``."""
        ch = ctypes.c_int64()
        ids = np.empty(self.m, dtype=np.int32) if want_ids else None
        N.check(self._lib.vbnmf_engine_cluster_changes(self._h, ctypes.byref(ch), ids.ctypes.data_as(N.c_int32_p) if want_ids else None))
        return (None if ch.value < 0 else ch.value), ids

    def random_state(self, hyper, seed):
        """vb_init(initializer='random') drawn on the device (reference R/bayesian.R:111-115); ``seed``: 64-bit."""
        N.check(self._lib.vbnmf_engine_random_state(self._h, float(hyper["aw"]), float(hyper["bw"]), float(hyper["ah"]),
                                                    float(hyper["bh"]), int(seed) & 0xFFFFFFFFFFFFFFFF))

    def svd(self, rank_out, tol=1e-7, maxit=60, seed=0):
        """Leading ``rank_out`` singular triplets of the resident X by device-resident subspace iteration on
        ``self.rank`` columns -> ``(u, d, vt, iterations)``."""
        u = np.empty((self.n, int(rank_out)), order="F")
        d = np.empty(int(rank_out))
        vt = np.empty((int(rank_out), self.m), order="F")
        it = ctypes.c_int32()
        N.check(self._lib.vbnmf_engine_svd(self._h, int(rank_out), float(tol), int(maxit), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                           N.dptr(u), N.dptr(d), N.dptr(vt), ctypes.byref(it)))
        return u, d, vt, it.value

    # -- sparse products with the resident X (truncated SVD of the svd2 initialiser) ----------
    def spmm(self, B, transpose=False):
        """``X @ B.T`` (B: r x m -> n x r) or, with ``transpose``, ``B.T @ X`` (B: n x r -> r x m); drops any state."""
        B = N.fcol(B)
        if not transpose:
            if B.shape != (self.rank, self.m):
                raise ValueError(f"B must be {(self.rank, self.m)}")
            out = np.empty((self.n, self.rank), order="F")
        else:
            if B.shape != (self.n, self.rank):
                raise ValueError(f"B must be {(self.n, self.rank)}")
            out = np.empty((self.rank, self.m), order="F")
        N.check(self._lib.vbnmf_engine_spmm(self._h, int(bool(transpose)), N.dptr(B), N.dptr(out)))
        return out

    def run(self, hyper, Itmax=10000, Tol=1e-5, n0=10, dn=1, flags=(True,) * 4, fudge=EPS, history=False):
        """The per-rank loop of vb_iterate (reference R/bayesian.R:336-352) driven by the device.

        Returns ``dict(it, lk0, lkh, hyper, reason, history)``; ``reason``: 1 NaN evidence, 2 converged,
        3 hyper-parameter Newton failed (raised as RuntimeError, as the reference stops there), 4 Itmax.
        ``history`` (optional): array [it, 9] = lkh, mean log lw, mean log lh, mean ew, mean eh, aw, bw, ah, bh."""
        hy = (ctypes.c_double * 4)(*(float(hyper[k]) for k in ("aw", "bw", "ah", "bh")))
        fl = (ctypes.c_int32 * 4)(*(1 if f else 0 for f in flags))
        it, reason = ctypes.c_int32(), ctypes.c_int32()
        lk0, lkh = ctypes.c_double(), ctypes.c_double()
        hist = np.zeros((int(Itmax), 9)) if history else None
        N.check(self._lib.vbnmf_engine_run(self._h, hy, float(fudge), int(Itmax), float(Tol), int(n0), int(dn), fl,
                                           ctypes.byref(it), ctypes.byref(lk0), ctypes.byref(lkh), ctypes.byref(reason),
                                           N.dptr(hist), int(Itmax) if history else 0))
        if reason.value == 3:
            raise RuntimeError("Hyper-parameter update failed to converge")      # reference R/bayesian.R:43
        return {"it": it.value, "lk0": lk0.value, "lkh": lkh.value, "reason": reason.value,
                "hyper": dict(zip(("aw", "bw", "ah", "bh"), (float(v) for v in hy))),
                "history": hist[:it.value] if history else None}

    # -- communicators (cell-partitioned runs) -------------------------------------------------
    def attach_comm(self, comm):
        N.check(self._lib.vbnmf_engine_attach_comm(self._h, comm._h))
        self.comm = comm

    def allreduce(self):
        """In-place RCCL all-reduce of the reduce buffer on the engine's stream (native communicator attached)."""
        N.check(self._lib.vbnmf_engine_allreduce(self._h))

    def step_local(self, hyper, fudge=EPS):
        N.check(self._lib.vbnmf_engine_step_local(self._h, hyper["aw"], hyper["bw"], hyper["ah"], hyper["bh"], float(fudge)))

    def step_finish(self):
        lkh = ctypes.c_double()
        st = (ctypes.c_double * 4)()
        N.check(self._lib.vbnmf_engine_step_finish(self._h, ctypes.byref(lkh), st))
        return lkh.value, tuple(st)

    def reduce_buffer(self):
        """(device pointer, count of doubles) of the buffer a partitioned run all-reduces."""
        p, c = ctypes.c_void_p(), ctypes.c_int64()
        N.check(self._lib.vbnmf_engine_reduce_buffer(self._h, ctypes.byref(p), ctypes.byref(c)))
        return p.value, c.value

    def reduce_tensor(self):
        """The reduce buffer as a torch CUDA tensor aliasing the engine's memory."""
        import torch
        p, c = self.reduce_buffer()
        return torch.as_tensor(_CudaBuffer(p, c), device=f"cuda:{self.device}")

    def stream(self):
        s = ctypes.c_void_p()
        N.check(self._lib.vbnmf_engine_get_stream(self._h, ctypes.byref(s)))
        return s.value

    def set_stream(self, stream_ptr: int):
        N.check(self._lib.vbnmf_engine_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    # -- instrumentation -------------------------------------------------------------------
    def timing_enable(self, on=True):
        N.check(self._lib.vbnmf_engine_timing_enable(self._h, 1 if on else 0))

    def timing_get(self):
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        N.check(self._lib.vbnmf_engine_timing_get(self._h, ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    def layout_info(self):
        v = [ctypes.c_int64() for _ in range(6)]
        N.check(self._lib.vbnmf_engine_layout_info(self._h, *[ctypes.byref(x) for x in v]))
        keys = ("nnz", "slots_gene_side", "slots_cell_side", "stream_bytes_per_step", "tasks_gene_side", "tasks_cell_side")
        return dict(zip(keys, (x.value for x in v)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.vbnmf_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def padded_rank(rank):
    """The row width the kernels store a factor of this rank in (``vbnmf_padded_rank``)."""
    return int(N.load().vbnmf_padded_rank(int(rank)))


def batch_grid(B):
    """Grid of the engines of a batch of B: 256 / B workgroups and blocks each (a multiple of 8 -- one per XCD --, at least 8)."""
    g = max(8, (256 // max(1, int(B))) // 8 * 8)
    return (g, g)


def auto_batch(nnz, nrun):
    """How many restarts of a rank the drivers step together when not told: the smaller the matrix, the more of them one launch
    holds with profit (profiles/r05_batch_sizes.txt, r05_batch_ml.txt: a batch's engines sit on 256 / B workgroups each, too
    few for a matrix that could use the chip by itself)."""
    cap = 16 if nnz <= 1_000_000 else 8 if nnz <= 4_000_000 else 4 if nnz <= 20_000_000 else 1
    return max(1, min(int(nrun), cap))


def run_batch(engines, hypers, Itmax=10000, Tol=1e-5, n0=10, dn=1, flags=(True,) * 4, fudge=EPS, history=False):
    """The device-driven loops of several engines of ONE rank on ONE ``CountMatrix`` -- the restarts of a rank, reference
    R/bayesian.R:260-261 -- stepped together (``vbnmf_batch_run``: two launches per step for the whole batch).  Every
    engine has had its state set; ``hypers`` holds one mapping per engine.  Returns one ``VBEngine.run`` result per engine,
    bit for bit what ``run`` gives on that engine alone; ``reason`` 3 (hyper-parameter Newton failed) is returned, not
    raised: the other engines' results are valid."""
    B = len(engines)
    if B < 1 or len(hypers) != B:
        raise ValueError("one hyper-parameter mapping per engine")
    lib = engines[0]._lib
    hs = (ctypes.c_void_p * B)(*(e._h for e in engines))
    hy = np.array([[float(h[k]) for k in ("aw", "bw", "ah", "bh")] for h in hypers], dtype=np.float64)
    fl = (ctypes.c_int32 * 4)(*(1 if f else 0 for f in flags))
    it = np.zeros(B, dtype=np.int32); reason = np.zeros(B, dtype=np.int32)
    lk0 = np.zeros(B); lkh = np.zeros(B)
    hist = np.zeros((B, int(Itmax), 9)) if history else None
    N.check(lib.vbnmf_batch_run(hs, B, N.dptr(hy), float(fudge), int(Itmax), float(Tol), int(n0), int(dn), fl,
                                it.ctypes.data_as(N.c_int32_p), N.dptr(lk0), N.dptr(lkh), reason.ctypes.data_as(N.c_int32_p),
                                N.dptr(hist), int(Itmax) if history else 0))
    return [{"it": int(it[b]), "lk0": float(lk0[b]), "lkh": float(lkh[b]), "reason": int(reason[b]),
             "hyper": dict(zip(("aw", "bw", "ah", "bh"), (float(v) for v in hy[b]))),
             "history": hist[b, :it[b]].copy() if history else None} for b in range(B)]


def run_batch_ml(engines, Itmax=10000, Tol=1e-5, prior=False, gamma_a=1.0, gamma_b=1.0, history=False):
    """The ML-NMF loops (``VBEngine.ml_run``) of several engines of ONE rank on ONE ``CountMatrix`` -- the ``nrun`` restarts
    ``factorize`` makes of a rank, reference R/factorize.R:181 -- stepped together (``vbnmf_batch_ml_run``).  Every engine has
    had ``ml_set_state``.  Returns one ``ml_run`` result per engine, bit for bit what ``ml_run`` gives on that engine alone."""
    B = len(engines)
    if B < 1:
        raise ValueError("an empty batch")
    lib = engines[0]._lib
    hs = (ctypes.c_void_p * B)(*(e._h for e in engines))
    it = np.zeros(B, dtype=np.int32); reason = np.zeros(B, dtype=np.int32)
    lk = np.zeros(B)
    hist = np.zeros((B, int(Itmax))) if history else None
    N.check(lib.vbnmf_batch_ml_run(hs, B, int(bool(prior)), float(gamma_a), float(gamma_b), int(Itmax), float(Tol),
                                   it.ctypes.data_as(N.c_int32_p), N.dptr(lk), reason.ctypes.data_as(N.c_int32_p),
                                   N.dptr(hist), int(Itmax) if history else 0))
    return [{"it": int(it[b]), "lk": float(lk[b]), "reason": int(reason[b]),
             "history": hist[b, :it[b]].copy() if history else None} for b in range(B)]


def _run_outputs(Itmax, history):
    return (ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double(), ctypes.c_double(),
            np.zeros((int(Itmax), 9)) if history else None)


class Communicator:
    """``vbnmf_comm``: the all-reduce of a cell-partitioned run, owned by the library.

    ``Communicator.rccl(id, nranks, rank, device)`` -- one process per GPU; ``id`` comes from
    ``Communicator.unique_id()`` on rank 0 and is broadcast by the caller.  ``Communicator.local(nranks, device)`` --
    the partition engines share this process and one device (tests, single-GPU rehearsals)."""

    def __init__(self, handle, kind, nranks, rank):
        self._lib = N.load()
        self._h = handle
        self.kind, self.nranks, self.rank = kind, nranks, rank

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(N.COMM_ID_BYTES)
        N.check(N.load().vbnmf_comm_unique_id(buf, N.COMM_ID_BYTES))
        return buf.raw

    @classmethod
    def rccl(cls, uid, nranks, rank, device):
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(uid), N.COMM_ID_BYTES)
        N.check(N.load().vbnmf_comm_create(buf, N.COMM_ID_BYTES, int(nranks), int(rank), int(device), ctypes.byref(h)))
        return cls(h, "rccl", int(nranks), int(rank))

    @classmethod
    def local(cls, nranks, device=0):
        h = ctypes.c_void_p()
        N.check(N.load().vbnmf_comm_create_local(int(nranks), int(device), ctypes.byref(h)))
        return cls(h, "local", int(nranks), 0)

    # -- local groups: every member engine attached, in partition order ---------------------------------
    def state_finish(self):
        N.check(self._lib.vbnmf_group_state_finish(self._h))

    def run(self, hyper, Itmax=10000, Tol=1e-5, n0=10, dn=1, flags=(True,) * 4, fudge=EPS, history=False):
        """``VBEngine.run`` for the whole local group (same result dictionary)."""
        hy = (ctypes.c_double * 4)(*(float(hyper[k]) for k in ("aw", "bw", "ah", "bh")))
        fl = (ctypes.c_int32 * 4)(*(1 if f else 0 for f in flags))
        it, reason, lk0, lkh, hist = _run_outputs(Itmax, history)
        N.check(self._lib.vbnmf_group_run(self._h, hy, float(fudge), int(Itmax), float(Tol), int(n0), int(dn), fl,
                                          ctypes.byref(it), ctypes.byref(lk0), ctypes.byref(lkh), ctypes.byref(reason),
                                          N.dptr(hist), int(Itmax) if history else 0))
        if reason.value == 3:
            raise RuntimeError("Hyper-parameter update failed to converge")      # reference R/bayesian.R:43
        return {"it": it.value, "lk0": lk0.value, "lkh": lkh.value, "reason": reason.value,
                "hyper": dict(zip(("aw", "bw", "ah", "bh"), (float(v) for v in hy))),
                "history": hist[:it.value] if history else None}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.vbnmf_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
