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
        out = {k: getattr(v, k) for k in ("side", "wide", "n_major", "n_minor", "block_width", "n_blocks", "chunk",
                                          "n_tiles", "n_slices", "n_slots")}
        arr = lambda p, cnt: np.ctypeslib.as_array(p, shape=(cnt,)).copy() if cnt else np.zeros(0)
        out["tile_block"] = arr(v.tile_block, v.n_tiles)
        out["tile_slice0"] = arr(v.tile_slice0, v.n_tiles + 1)
        out["slice_major"] = arr(v.slice_major, v.n_slices * 64)
        out["slice_width"] = arr(v.slice_width, v.n_slices)
        out["slice_off"] = arr(v.slice_off, v.n_slices)
        if v.wide:
            out["wide_idx"] = arr(v.wide_idx, v.n_slots)
            out["wide_val"] = arr(v.wide_val, v.n_slots)
        else:
            out["packed"] = arr(v.packed, v.n_slots)
    finally:
        L.vbnmf_layout_destroy(h)
    return out


def reconstruct(view):
    """Dense [n_major, n_minor] matrix the layout encodes, plus the list of (major, block) pairs seen."""
    A = np.zeros((view["n_major"], view["n_minor"]))
    seen = np.zeros((view["n_major"], view["n_blocks"]), dtype=np.int64)
    C = view["block_width"]
    for t in range(view["n_tiles"]):
        blk = view["tile_block"][t]
        for s in range(view["tile_slice0"][t], view["tile_slice0"][t + 1]):
            w, off = view["slice_width"][s], view["slice_off"][s]
            assert w % 4 == 0 and off % 256 == 0
            for lane in range(64):
                M = view["slice_major"][s * 64 + lane]
                tt = np.arange(w)
                slots = off + (tt // 4) * 256 + lane * 4 + tt % 4
                if view["wide"]:
                    idx, val = view["wide_idx"][slots], view["wide_val"][slots]
                else:
                    e = view["packed"][slots]
                    idx, val = e & 0xFFFF, (e >> 16).astype(np.float64)
                if M == 0xFFFFFFFF:
                    assert not val.any()
                    continue
                seen[M, blk] += 1
                live = val != 0
                assert (idx[~live] == 0).all()
                cols = blk * C + idx[live]
                assert (cols < min((blk + 1) * C, view["n_minor"])).all()
                assert np.all(np.diff(cols) > 0)          # minors ascending within a lane
                assert not live[np.argmin(live):].any() if not live.all() else True   # padding only at the tail
                A[M, cols] += val[live]
    return A, seen
