#!/usr/bin/env python3
"""tests/manual_asan_host.py -- AddressSanitizer / UBSan pass over the HOST side of the library (ingestion, tiled
layout, Matrix Market reader / writer); GPU sanitizers are not available on the pool, so this is the CPU build only.

    g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer \\
        -o /tmp/libhost_asan.so ccfindr_amd/csrc/host.cpp ccfindr_amd/csrc/mtx.cpp -pthread
    LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \\
        python3 tests/manual_asan_host.py /tmp/libhost_asan.so

40 random matrices (dense / shuffled CSC / CSR ingestion; integer, non-integer and > 16383 values), both sides at
ranks 1..32, whole and partial column ranges, Matrix Market round trips and malformed files.  Last run: clean.
"""
import ctypes, sys, os, numpy as np, scipy.sparse as sp, tempfile
L = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else '/tmp/libhost_asan.so')
i64, i32, vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
def ptr(a): return a.ctypes.data_as(vp)
rng = np.random.default_rng(0)
view = (ctypes.c_char * 1024)()
for trial in range(40):
    n, m = int(rng.integers(1, 300)), int(rng.integers(1, 500))
    X = rng.poisson(rng.choice([0.02, 0.3, 2.0]), size=(n, m)).astype(np.float64)
    kind = trial % 4
    if kind == 1: X *= rng.uniform(0.5, 1.5, size=(1, m))
    if kind == 2 and X.size: X[rng.integers(0, n), rng.integers(0, m)] = 40000 + trial
    if not X.any(): X[0, 0] = 1
    h = vp()
    if trial % 3 == 0:
        A = np.asfortranarray(X); rc = L.vbnmf_matrix_from_dense(i64(n), i64(m), ptr(A), ctypes.byref(h))
    elif trial % 3 == 1:
        S = sp.csc_matrix(X); p = S.indptr.astype(np.int32); i = S.indices.astype(np.int32); x = S.data.astype(np.float64)
        perm = np.concatenate([rng.permutation(np.arange(p[j], p[j+1])) for j in range(m)]).astype(np.int64) if S.nnz else np.zeros(0, dtype=np.int64)
        i2, x2 = i[perm].copy(), x[perm].copy()
        rc = L.vbnmf_matrix_from_csc(i64(n), i64(m), ptr(p), ptr(i2), ptr(x2), ctypes.byref(h))
    else:
        S = sp.csr_matrix(X); p = S.indptr.astype(np.int32); j = S.indices.astype(np.int32); x = S.data.astype(np.float64)
        rc = L.vbnmf_matrix_from_csr(i64(n), i64(m), ptr(p), ptr(j), ptr(x), ctypes.byref(h))
    assert rc == 0, rc
    for side in (0, 1):
        for r in (1, 6, 10, 20, 32):
            lh = vp()
            cb, ce = (0, m) if trial % 5 else (m // 3, max(m // 3 + 1, 2 * m // 3))
            rc = L.vbnmf_layout_build(h, i64(cb), i64(ce), i32(side), i32(r), ctypes.byref(lh), view)
            assert rc == 0, (rc, n, m, side, r)
            L.vbnmf_layout_destroy(lh)
    with tempfile.TemporaryDirectory() as t:
        path = os.path.join(t, "a.mtx").encode()
        assert L.vbnmf_matrix_write_mtx(h, path) == 0
        h2 = vp(); assert L.vbnmf_matrix_from_mtx(path, ctypes.byref(h2)) == 0
        L.vbnmf_matrix_destroy(h2)
    L.vbnmf_matrix_destroy(h)
# malformed files
for txt in [b"", b"%%MatrixMarket matrix coordinate integer general\n", b"%%MatrixMarket matrix coordinate integer general\n2 2 5\n1 1 1\n", b"%%MatrixMarket matrix coordinate real general\n2 2 1\n9 9 1\n", b"junk\n1 1 1\n", b"%%MatrixMarket matrix array real general\n2 2\n1\n2\n"]:
    with tempfile.NamedTemporaryFile(suffix=".mtx") as f:
        f.write(txt); f.flush(); h = vp()
        rc = L.vbnmf_matrix_from_mtx(f.name.encode(), ctypes.byref(h)); assert rc != 0 or txt == b""
print("asan drive ok")
