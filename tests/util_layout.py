"""Helpers for tests: read the tiled device layout back through the C ABI (host only)."""
import ctypes

import numpy as np

from ccfindr_amd import _native as N


def build_layout(M, side, r, cols=None):
    """Returns (view dict of numpy copies, None).  M: ccfindr_amd.CountMatrix."""
    L = N.load()
    cb, ce = cols if cols is not None else (0, M.shape[1])
    h = ctypes.c_void_p()
    v = N.LayoutView()
    N.check(L.vbnmf_layout_build(M._h, cb, ce, side, r, ctypes.byref(h), ctypes.byref(v)))
    try:
        out = {k: getattr(v, k) for k in ("side", "wide", "n_major", "n_minor", "block_width", "n_blocks", "max_len",
                                          "n_wg", "row_slots", "n_tasks", "n_slices", "n_slots", "n_segs")}
        arr = lambda p, cnt: np.ctypeslib.as_array(p, shape=(cnt,)).copy() if cnt else np.zeros(0, dtype=np.int64)
        out["task_major"] = arr(v.task_major, v.n_slices * 64)
        out["slice_width"] = arr(v.slice_width, v.n_slices)
        out["slice_off"] = arr(v.slice_off, v.n_slices)
        out["slice_block"] = arr(v.slice_block, v.n_slices)
        raw = arr(v.slice_fast, v.n_slices)
        out["slice_fast"] = raw & 0xFFFF                      # the leading stretch of ones ...
        out["slice_fast2"] = raw >> 16                        # ... and of ones or twos (>= it)
        out["seg_block"] = arr(v.seg_block, v.n_segs)
        out["block_start"] = arr(v.block_start, v.n_blocks + 1)
        out["seg_ptr"] = arr(v.seg_ptr, v.n_segs + 1)
        out["wg_seg0"] = arr(v.wg_seg0, v.n_wg + 1)
        out["inv_ptr"] = arr(v.inv_ptr, v.n_major + 1)
        out["inv_task"] = arr(v.inv_task, v.n_tasks)
        ncell = v.n_minor if v.side == 0 else v.n_major
        out["cell_perm"] = arr(v.cell_perm, ncell) if v.cell_perm else None     # position -> column (None: as stored)
        if v.wide:
            out["wide_idx"] = arr(v.wide_idx, v.n_slots)
            out["wide_val"] = arr(v.wide_val, v.n_slots)
        else:
            out["packed"] = arr(v.packed, v.n_slots)
    finally:
        L.vbnmf_layout_destroy(h)
    return out


def reconstruct(view):
    """Dense [n_major, n_minor] matrix the layout encodes, in X's own numbering of the cells (the layout's internal
    renumbering, view["cell_perm"], is undone at the end); checks the structural invariants on the way."""
    A = np.zeros((view["n_major"], view["n_minor"]))
    nslot = np.zeros(A.shape, dtype=np.int64)          # slots per (major, minor)
    npart = np.zeros(A.shape, dtype=np.int64)          # ... of which not a full 16383 piece
    bstart = view["block_start"]
    assert bstart[0] == 0 and bstart[-1] == view["n_minor"] and np.all(np.diff(bstart) > 0)
    assert np.max(np.diff(bstart)) == view["block_width"]          # block_width = the widest block (what the LDS is sized for)
    ntask = 0
    first_minor = {}
    for s in range(view["n_slices"]):
        blk = view["slice_block"][s]
        w, off = view["slice_width"][s], view["slice_off"][s]
        assert w % 4 == 0 and w >= 4 and w <= view["max_len"] and off % 256 == 0
        lens = []
        ones = []                                                  # per lane: its leading stretch of stored ones
        ones12 = []                                                # per lane: ones and twos
        for lane in range(64):
            tid = s * 64 + lane
            M = view["task_major"][tid]
            tt = np.arange(w)
            slots = off + (tt // 4) * 256 + lane * 4 + tt % 4
            if view["wide"]:
                idx, val = view["wide_idx"][slots], view["wide_val"][slots]
            else:
                e = view["packed"][slots]
                idx, val = ((e >> 4) & 0x3FFF) // view["row_slots"], (e >> 18).astype(np.float64)
            if M == 0xFFFFFFFF:
                assert not val.any() and not idx.any()
                lens.append(0)
                ones.append(0)
                continue
            ntask += 1
            live = val != 0
            n_live = int(live.sum())
            assert n_live >= 1 and live[:n_live].all()        # entries first, padding only at the tail
            assert (idx[~live] == 0).all()
            cols = bstart[blk] + idx[live]
            assert (cols < bstart[blk + 1]).all()
            # any order (bank-conflict schedule).  A (major, minor) pair occupies one slot, unless a count above the
            # packed range was split: then all but one of its slots hold a full piece (16383) -- checked at the end.
            np.add.at(A, (np.full(cols.size, M), cols), val[live])
            np.add.at(nslot, (np.full(cols.size, M), cols), 1)
            np.add.at(npart, (np.full(cols.size, M), cols), (val[live] != 16383).astype(np.int64))
            first_minor[tid] = int(cols.min())
            lens.append(n_live)
            if view["wide"]:
                ones.append(0)
            else:
                is1 = val[:n_live] == 1.0                          # a task's ones come first, its other entries after them
                n1 = int(is1.sum())
                assert is1[:n1].all()
                ones.append(n1)
        padded = [(q + 3) // 4 for q in lens]
        assert padded == sorted(padded, reverse=True)          # longest (padded) task first: width = first lane
        assert (w - lens[0]) < 4
        fast = int(view["slice_fast"][s])
        assert fast % 8 == 0 and 0 <= fast <= min(ones) and (view["wide"] == 0 or fast == 0)
    assert ntask == view["n_tasks"]
    multi = nslot > 1
    assert (npart[multi] <= 1).all() and (not multi.any() or not view["wide"])      # no entry stored twice
    # inverse index: every task exactly once, under its own major, in ascending minor order
    seen = np.zeros(view["n_slices"] * 64, dtype=np.int64)
    for M in range(view["n_major"]):
        ids = view["inv_task"][view["inv_ptr"][M]:view["inv_ptr"][M + 1]]
        seen[ids] += 1
        assert (view["task_major"][ids] == M).all()
        fm = [first_minor[int(t)] for t in ids]
        assert fm == sorted(fm)
    assert (seen[view["task_major"] != 0xFFFFFFFF] == 1).all() and seen.sum() == view["n_tasks"]
    # persistent workgroups: the segments tile the slice ids in order; one block per segment; inside a segment
    # the slices (pulled through the ticket counter in id order) go longest first
    assert view["wg_seg0"][0] == 0 and view["wg_seg0"][-1] == view["n_segs"]
    assert np.all(np.diff(view["wg_seg0"]) >= 0)
    ptr = view["seg_ptr"]
    assert ptr[0] == 0 and ptr[-1] == view["n_slices"] and np.all(np.diff(ptr) > 0)
    for g in range(view["n_segs"]):
        assert (view["slice_block"][ptr[g]:ptr[g + 1]] == view["seg_block"][g]).all()
        assert np.all(np.diff(view["slice_width"][ptr[g]:ptr[g + 1]]) <= 0)
    perm = view.get("cell_perm")
    if perm is not None:
        assert sorted(perm.tolist()) == list(range(perm.size))          # a permutation of the cells
        B = np.empty_like(A)
        if view["side"] == 0:
            B[:, perm] = A                                              # minor position p holds column perm[p]
        else:
            B[perm, :] = A
        A = B
    return A


def workgroup_balance(view, c0=10):
    """Cost (entries per lane + c0 per slice) of every workgroup's share: how evenly the persistent workgroups are loaded."""
    ptr = view["seg_ptr"]
    w = view["slice_width"].astype(np.int64) + c0
    return np.array([w[ptr[view["wg_seg0"][g]]:ptr[view["wg_seg0"][g + 1]]].sum() for g in range(view["n_wg"])])
