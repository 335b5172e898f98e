"""The tiled device layout holds exactly X (bit-exact integer/byte work; host only, no GPU)."""
import numpy as np
import pytest
import scipy.sparse as sp

from util_layout import build_layout, reconstruct


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
        A, seen = reconstruct(v)
        assert np.array_equal(A, X if side == 0 else X.T)
        assert (seen == 1).all()                 # every (major, block) pair is owned by exactly one lane
        assert v["block_width"] * 8 * ((r + 1) // 2 * 2) <= 160 * 1024


def test_layout_roundtrip_wide_and_partition():
    import ccfindr_amd as C
    X = _counts(120, 200, 0.4, seed=9) * 0.37
    M = C.CountMatrix(sp.csc_matrix(X))
    for side in (0, 1):
        v = build_layout(M, side, 4, cols=(50, 170))
        assert v["wide"] == 1
        A, seen = reconstruct(v)
        Xs = X[:, 50:170]
        assert np.array_equal(A, Xs if side == 0 else Xs.T)
        assert (seen == 1).all()


def test_many_blocks_small_lds(monkeypatch):
    """Force narrow minor blocks so tiles span several blocks and chunks."""
    import ccfindr_amd as C
    monkeypatch.setenv("VBNMF_LDS_KB", "8")
    monkeypatch.setenv("VBNMF_CHUNK", "64")
    X = _counts(200, 700, 0.2, seed=3)
    M = C.CountMatrix(X)
    for side in (0, 1):
        v = build_layout(M, side, 6)
        assert v["n_blocks"] > 1 and v["n_tiles"] > v["n_blocks"]
        A, seen = reconstruct(v)
        assert np.array_equal(A, X if side == 0 else X.T)
        assert (seen == 1).all()
        # heaviest tiles first
        slots = [int((v["slice_width"][v["tile_slice0"][t]:v["tile_slice0"][t + 1]]).sum()) for t in range(v["n_tiles"])]
        assert slots == sorted(slots, reverse=True)


def test_ingestion_canonicalises_unsorted_duplicates_and_zeros():
    import ccfindr_amd as C
    # column 0: rows (3, 1, 3) with values (2, 5, 4) -> row1=5, row3=6 ; an explicit zero is dropped
    p = np.array([0, 3, 4, 4], dtype=np.int32)
    i = np.array([3, 1, 3, 2], dtype=np.int32)
    x = np.array([2.0, 5.0, 4.0, 0.0])
    M = C.CountMatrix.from_csc(5, 3, p, i, x)
    assert M.nnz == 2
    v = build_layout(M, 0, 2)
    A, _ = reconstruct(v)
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
