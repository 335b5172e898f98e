"""The tiled device layout holds exactly X (bit-exact integer/byte work; host only, no GPU)."""
import numpy as np
import pytest
import scipy.sparse as sp

from util_layout import build_layout, reconstruct, workgroup_balance


def _counts(n, m, lam, seed):
    rng = np.random.default_rng(seed)
    return np.asfortranarray(rng.poisson(lam, size=(n, m)).astype(np.float64))


@pytest.mark.parametrize("n,m,r,lam", [(50, 70, 3, 0.5), (300, 130, 10, 0.1), (1100, 90, 2, 0.3), (64, 2500, 20, 0.05)])
def test_layout_roundtrip_packed(n, m, r, lam):
    import ccfindr_amd as C
    X = _counts(n, m, lam, seed=n + m)
    M = C.CountMatrix(X)
    for side in (0, 1):
        v = build_layout(M, side, r)
        assert v["wide"] == 0
        A = reconstruct(v)
        assert np.array_equal(A, X if side == 0 else X.T)
        R = (r + 1) // 2 * 2
        assert v["block_width"] * (((R // 2) | 1) * 16) <= 160 * 1024
        assert v["n_slots"] >= M.nnz


def test_layout_roundtrip_wide_and_partition():
    import ccfindr_amd as C
    X = _counts(120, 200, 0.4, seed=9) * 0.37
    M = C.CountMatrix(sp.csc_matrix(X))
    for side in (0, 1):
        v = build_layout(M, side, 4, cols=(50, 170))
        assert v["wide"] == 1
        A = reconstruct(v)
        Xs = X[:, 50:170]
        assert np.array_equal(A, Xs if side == 0 else Xs.T)


def test_long_rows_are_split_and_blocks_are_narrow(monkeypatch):
    """Force narrow minor blocks, short tasks and few workgroups: rows split into several tasks,
    several segments per workgroup."""
    import ccfindr_amd as C
    monkeypatch.setenv("VBNMF_LDS_KB", "8")
    monkeypatch.setenv("VBNMF_MAX_LEN", "8")
    monkeypatch.setenv("VBNMF_NWG", "3")
    X = _counts(200, 700, 0.3, seed=3)
    X[5, :] = 2.0                                  # a dense gene
    M = C.CountMatrix(X)
    for side in (0, 1):
        v = build_layout(M, side, 6)
        assert v["n_blocks"] > 1 and v["max_len"] == 8 and v["n_wg"] == 3
        assert v["n_segs"] >= v["n_blocks"]
        A = reconstruct(v)
        assert np.array_equal(A, X if side == 0 else X.T)
        per_major = np.diff(v["inv_ptr"])
        if side == 0:
            assert per_major[5] > v["n_blocks"]     # the dense gene needed more than one task per block


def test_padding_is_small_on_skewed_data():
    """Genes differ hugely in expression; sorting tasks by length keeps slot padding low."""
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.fill_empty(synth.simulate_data(3000, [2000] * 3, alpha0=0.065, seed=5, depth=np.full(6000, 300)))
    M = C.CountMatrix(X)
    for side in (0, 1):
        v = build_layout(M, side, 10)
        assert v["n_slots"] <= 1.12 * M.nnz, (side, v["n_slots"] / M.nnz)


def test_workgroup_shares_are_balanced(monkeypatch):
    """Whole workgroups per block by cost, a block's slices snake-dealt: no share far above the mean."""
    import ccfindr_amd as C
    monkeypatch.setenv("VBNMF_NWG", "16")          # ~11 slices per workgroup on this small matrix
    from ccfindr_amd import synth
    X = synth.fill_empty(synth.simulate_data(3000, [2000] * 3, alpha0=0.065, seed=5, depth=np.full(6000, 300)))
    M = C.CountMatrix(X)
    for side in (0, 1):
        cost = workgroup_balance(build_layout(M, side, 10))
        assert cost.min() > 0 and cost.max() <= 1.25 * cost.mean(), (side, cost.max() / cost.mean())


def test_ingestion_canonicalises_unsorted_duplicates_and_zeros():
    import ccfindr_amd as C
    # column 0: rows (3, 1, 3) with values (2, 5, 4) -> row1=5, row3=6 ; an explicit zero is dropped
    p = np.array([0, 3, 4, 4], dtype=np.int32)
    i = np.array([3, 1, 3, 2], dtype=np.int32)
    x = np.array([2.0, 5.0, 4.0, 0.0])
    M = C.CountMatrix.from_csc(5, 3, p, i, x)
    assert M.nnz == 2
    v = build_layout(M, 0, 2)
    A = reconstruct(v)
    want = np.zeros((5, 3)); want[1, 0] = 5; want[3, 0] = 6
    assert np.array_equal(A, want)
    assert M.empty_counts() == (3, 2)


def test_bad_indices_are_rejected():
    import ccfindr_amd as C
    p = np.array([0, 1], dtype=np.int32)
    with pytest.raises(C.VBNMFError):
        C.CountMatrix.from_csc(3, 1, p, np.array([3], dtype=np.int32), np.array([1.0]))
    with pytest.raises(C.VBNMFError):
        C.CountMatrix.from_csc(3, 1, np.array([0, 2], dtype=np.int32), np.array([0, -1], dtype=np.int32), np.array([1.0, 1.0]))


def test_counts_above_the_packed_range_are_split_not_widened():
    """Integer counts beyond 16383 keep the 4-byte entry format: the entry becomes several slots of the same minor."""
    import ccfindr_amd as C
    rng = np.random.default_rng(11)
    X = rng.poisson(0.6, size=(90, 140)).astype(np.float64)
    X[np.arange(90), rng.integers(0, 140, 90)] += 1
    X[rng.integers(0, 90, 140), np.arange(140)] += 1
    X[5, 7] = 16383.0; X[6, 8] = 16384.0; X[7, 9] = 70000.0; X[8, 10] = 3 * 16383.0
    M = C.CountMatrix(X)
    for side in (0, 1):
        v = build_layout(M, side, 6)
        assert not v["wide"]
        A = reconstruct(v)
        assert np.array_equal(A, X if side == 0 else X.T)
    Y = X.copy(); Y[1, 1] = 2.5                       # one non-integer value: the wide layout
    assert build_layout(C.CountMatrix(Y), 0, 6)["wide"]


def test_rank_classes_share_one_geometry():
    """vbnmf_matrix_plan_ranks: every rank of a planned sweep takes the LDS row stride and block width of its class (the
    largest rank by default), so the layouts -- rank-independent otherwise -- are cut once; the layout still holds X."""
    import ccfindr_amd as C
    X = _counts(300, 900, 0.2, seed=21)
    M = C.CountMatrix(X)
    own = {r: build_layout(M, 0, r) for r in (3, 10, 20)}
    assert [own[r]["row_slots"] for r in (3, 10, 20)] == [3, 5, 11]
    M.plan_ranks([3, 10, 20])
    for side in (0, 1):
        views = [build_layout(M, side, r) for r in (3, 10, 20)]
        for v in views:
            assert v["row_slots"] == 11 and v["block_width"] * 11 * 16 <= 160 * 1024 - 4112
        for k in ("packed", "task_major", "slice_width", "slice_off", "inv_task", "seg_ptr", "wg_seg0"):
            assert all(np.array_equal(views[0][k], v[k]) for v in views[1:]), k
        assert np.array_equal(reconstruct(views[0]), X if side == 0 else X.T)
    M.plan_ranks([3, 10, 20], max_classes=2)                 # a second class where the rows are at most half as wide
    assert [build_layout(M, 0, r)["row_slots"] for r in (3, 10, 20)] == [5, 5, 11]
    M.plan_ranks(())                                         # cleared: every rank its own geometry again
    assert build_layout(M, 0, 3)["row_slots"] == 3
    M.close()
