"""CPU: the native Matrix Market reader / writer (ccfindr_amd/csrc/mtx.cpp) and the
read_10x / write_10x mirrors (reference R/utils.R:28-54, :867-884).  The checker is scipy.io.mmread, an
independent implementation of the same published format, plus the reference's bundled PBMC sample, whose counts
are committed as data in tests/golden/pbmc_extdata_r5.npz."""
import os

import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def native(path):
    import ccfindr_amd as C
    M = C.CountMatrix.from_mtx(str(path))
    try:
        return M.to_scipy()
    finally:
        M.close()


def same(a, b):
    a = sp.csc_matrix(a); b = sp.csc_matrix(b)
    a.sum_duplicates(); b.sum_duplicates(); a.eliminate_zeros(); b.eliminate_zeros()
    a.sort_indices(); b.sort_indices()
    return a.shape == b.shape and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data)


CASES = {
    "integer_general": "%%MatrixMarket matrix coordinate integer general\n% a comment\n\n4 5 6\n1 1 3\n4 5 7\n2 3 1\n2 1 9\n3 3 2\n1 5 4\n",
    "real_exponents": "%%MatrixMarket matrix coordinate real general\n3 3 4\n1 1 1.5e0\n2 2 -2.25E-1\n3 1 4.0\n3 3 1e3\n",
    "pattern": "%%MatrixMarket matrix coordinate pattern general\n3 4 3\n1 2\n3 4\n2 2\n",
    "symmetric": "%%MatrixMarket matrix coordinate integer symmetric\n4 4 4\n1 1 5\n3 1 2\n4 2 7\n4 4 1\n",
    "skew": "%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 1.5\n3 2 -4\n",
    "array_general": "%%MatrixMarket matrix array real general\n3 2\n1\n0\n2.5\n0\n-3\n4\n",
    "array_symmetric": "%%MatrixMarket matrix array integer symmetric\n3 3\n1\n2\n3\n4\n5\n6\n",
    "crlf_no_trailing_newline": "%%MatrixMarket matrix coordinate integer general\r\n2 2 2\r\n1 1 1\r\n2 2 5",
    "mixed_case_banner": "%%MatrixMarket MATRIX Coordinate Integer General\n2 3 1\n2 3 8\n",
    "tabs_and_spaces": "%%MatrixMarket matrix coordinate integer general\n  3\t3   2\n 1\t2\t 6 \n3 3 1\n",
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_reader_matches_scipy(tmp_path, name):
    p = tmp_path / f"{name}.mtx"
    p.write_bytes(CASES[name].encode())
    assert same(native(p), scipy.io.mmread(str(p)))


def test_duplicates_are_summed_and_zeros_dropped(tmp_path):
    p = tmp_path / "d.mtx"
    p.write_text("%%MatrixMarket matrix coordinate integer general\n3 3 6\n1 1 2\n1 1 3\n2 2 0\n3 1 4\n3 1 -4\n2 3 1\n")
    X = native(p)
    assert X.nnz == 2 and X[0, 0] == 5 and X[1, 2] == 1          # (3,1) cancels to an unstored zero


@pytest.mark.parametrize("text,msg", [
    ("%MatrixMarket matrix coordinate integer general\n1 1 1\n1 1 1\n", "banner"),
    ("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1 0\n", "field"),
    ("%%MatrixMarket matrix coordinate integer hermitian\n1 1 1\n1 1 1\n", "symmetry"),
    ("%%MatrixMarket matrix coordinate integer general\n2 2 3\n1 1 1\n2 2 1\n", "entry lines"),
    ("%%MatrixMarket matrix coordinate integer general\n2 2 1\n3 1 1\n", "outside"),
    ("%%MatrixMarket matrix coordinate integer general\n2 2 1\n1 x 1\n", "malformed"),
    ("%%MatrixMarket matrix coordinate integer general\n", "size line"),
])
def test_reader_errors(tmp_path, text, msg):
    import ccfindr_amd as C
    p = tmp_path / "bad.mtx"
    p.write_text(text)
    with pytest.raises(C.VBNMFError, match=msg):
        C.CountMatrix.from_mtx(str(p))


def test_missing_file():
    import ccfindr_amd as C
    with pytest.raises(C.VBNMFError, match="does not exist"):
        C.CountMatrix.from_mtx("/nonexistent/matrix.mtx")


def test_large_file_parsed_by_many_threads(tmp_path):
    rng = np.random.default_rng(5)
    n, m, nnz = 3000, 5000, 1_200_000
    i, j = rng.integers(1, n + 1, nnz), rng.integers(1, m + 1, nnz)
    v = rng.integers(1, 50, nnz)
    p = tmp_path / "big.mtx"
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate integer general\n%\n")
        f.write(f"{n} {m} {nnz}\n")
        f.write("\n".join(f"{a} {b} {c}" for a, b, c in zip(i, j, v)))
        f.write("\n")
    want = sp.coo_matrix((v.astype(np.float64), (i - 1, j - 1)), shape=(n, m)).tocsc()
    assert same(native(p), want)


def test_pbmc_sample_and_round_trip(tmp_path):
    """The reference's bundled 10x sample (counts from the committed fixture): file -> engine matrix -> file."""
    import ccfindr_amd as C
    from ccfindr_amd import io
    d = np.load(os.path.join(GOLD, "pbmc_extdata_r5.npz"))
    n, m = int(d["n"]), int(d["m"])
    X = sp.csc_matrix((d["data"].astype(np.float64), d["indices"], d["indptr"]), shape=(n, m))
    genes = [[f"ENSG{i:011d}", f"G{i}"] for i in range(n)]
    cells = [[f"CELL{j}-1"] for j in range(m)]
    x = io.CountData(X, genes, cells)
    io.write_10x(x, str(tmp_path))
    head = open(tmp_path / "matrix.mtx").read().split("\n")[:3]
    assert head[0] == "%%MatrixMarket matrix coordinate integer general" and head[1] == f"{n} {m} {X.nnz}"
    assert same(scipy.io.mmread(str(tmp_path / "matrix.mtx")), X)          # the writer, read by scipy
    y = io.read_10x(str(tmp_path))
    assert same(y.counts, X) and y.genes == genes and y.barcodes == cells and y.rownames[3] == genes[3][0]
    M = y.count_matrix()
    assert M.shape == (n, m) and M.nnz == X.nnz
    M.close()


def test_read_10x_guards_and_remove_zeros(tmp_path):
    from ccfindr_amd import io
    with pytest.raises(FileNotFoundError, match="Input directory"):
        io.read_10x(str(tmp_path / "nope"))
    with pytest.raises(FileNotFoundError, match="Count file"):
        io.read_10x(str(tmp_path))
    X = sp.csc_matrix(np.array([[1.0, 0, 2], [0, 0, 0], [3, 0, 0]]))
    x = io.CountData(X, [["g1"], ["g2"], ["g3"]], [["c1"], ["c2"], ["c3"]])
    io.write_10x(x, str(tmp_path))
    kept = io.read_10x(str(tmp_path))
    assert kept.counts.shape == (2, 2) and kept.rownames == ["g1", "g3"] and kept.colnames == ["c1", "c3"]
    full = io.read_10x(str(tmp_path), remove_zeros_=False)
    assert full.counts.shape == (3, 3) and same(full.counts, X)
    os.remove(tmp_path / "genes.tsv")
    with pytest.raises(FileNotFoundError, match="genes.tsv"):
        io.read_10x(str(tmp_path))


def test_noninteger_values_round_trip_exactly(tmp_path):
    import ccfindr_amd as C
    rng = np.random.default_rng(2)
    A = rng.uniform(size=(6, 7)) * (rng.uniform(size=(6, 7)) < 0.4)
    M = C.CountMatrix(A)
    M.write_mtx(str(tmp_path / "r.mtx"))
    M.close()
    assert open(tmp_path / "r.mtx").readline().strip() == "%%MatrixMarket matrix coordinate real general"
    assert same(native(tmp_path / "r.mtx"), A)
