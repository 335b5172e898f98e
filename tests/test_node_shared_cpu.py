"""One ingestion and one pair of layouts per NODE (no GPU needed for the host side): the layout blob round trip through a
/dev/shm segment into a matrix SHELL, the shell's refusals, the rank classes as a pure function, and the stateless
cache's two independent content hashes.  Reference behaviour replaced: the whole bundle shipped to every MPI slave,
/root/reference R/bayesian.R:252-263."""
import ctypes
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _matrix(seed=4, wide=False):
    from ccfindr_amd import synth
    X = synth.fill_empty(synth.simulate_data(700, [400, 500, 300], alpha0=0.2, seed=seed, depth=np.full(1200, 90)), seed=seed)
    if wide:
        X = X.astype(np.float64) * 0.37
    return X


def _layout_view(M, side, r):
    from ccfindr_amd import _native as N
    L = N.load()
    h, v = ctypes.c_void_p(), N.LayoutView()
    N.check(L.vbnmf_layout_build(M._h, 0, M.shape[1], side, r, ctypes.byref(h), ctypes.byref(v)))
    return h, v


@pytest.mark.parametrize("wide", [False, True])
def test_layout_blob_round_trip_through_shared_memory(wide):
    """export -> /dev/shm segment -> import into a shell: the imported layout is accepted in place of a cut one, and a blob
    exported again FROM the shell is byte-identical to the original (every array and scalar survived)."""
    import ccfindr_amd as C
    from ccfindr_amd import node
    M = C.CountMatrix(_matrix(wide=wide))
    meta = M.meta()
    assert meta[0] == M.shape[0] and meta[1] == M.shape[1] and meta[2] == M.nnz and meta[6] == M.sum_lgamma_x1
    S = C.CountMatrix.shell(meta)
    assert S.is_shell and not M.is_shell and S.shape == M.shape and S.nnz == M.nnz
    for side in (0, 1):
        nb = M.layout_blob_size(side, 12, 256)
        seg = node.Segment.create(node.fresh_name("t"), nb)
        try:
            assert M.export_layout(side, 12, 256, seg.map) == nb
            peer = node.Segment.open(seg.name)
            S.import_layout(peer.map, nb)
            back = bytearray(nb)
            assert S.export_layout(side, 12, 256, back) == nb          # from the shell's cache: no entries needed
            assert bytes(back) == bytes(seg.map[:nb])
            peer.close()
        finally:
            seg.close()
        assert not os.path.exists(os.path.join(node.shm_dir(), seg.name))
    S.close(); M.close()


def test_shell_refuses_what_needs_entries_and_unknown_geometries():
    import ccfindr_amd as C
    from ccfindr_amd import _native as N
    M = C.CountMatrix(_matrix())
    S = C.CountMatrix.shell(M.meta())
    for call in (S.empty_counts, S.to_scipy, lambda: S.write_mtx("/tmp/never.mtx"), lambda: _layout_view(S, 0, 4),
                 lambda: S.layout_blob_size(0, 6, 256)):                # nothing imported for that geometry
        with pytest.raises(N.VBNMFError) as ei:
            call()
        assert ei.value.code == N.ERR_STATE and "shell" in str(ei.value)
    # a blob of another matrix, a truncated blob and a corrupted closing word are all refused
    nb = M.layout_blob_size(1, 6, 256)
    blob = bytearray(nb)
    M.export_layout(1, 6, 256, blob)
    other = C.CountMatrix.shell(np.array([M.shape[0], M.shape[1] + 1, M.nnz, 1, 1, 5, 0, 0], dtype=float))
    for target, data in ((other, blob), (S, blob[:nb - 64]), (S, blob[:nb - 8] + b"\0" * 8)):
        with pytest.raises(N.VBNMFError) as ei:
            target.import_layout(data)
        assert ei.value.code == N.ERR_BAD_ARG
    S.import_layout(blob)                                               # the intact one goes in
    assert S.layout_blob_size(1, 6, 256) == nb
    other.close(); S.close(); M.close()


def test_rank_classes_are_a_pure_function_and_match_the_matrix_plan():
    """vbnmf_plan_classes / vbnmf_padded_rank: what vb_factorize hands to each engine instead of mutating the matrix."""
    import ccfindr_amd as C
    from ccfindr_amd.engine import geometry_rank_for, rank_classes
    assert rank_classes(range(2, 21)) == [20]
    assert rank_classes([3, 10, 20], 2) == [10, 20]                     # rows of rank 10 are at most half as wide as 20's
    assert rank_classes([5]) == [6] and rank_classes([]) == []
    assert rank_classes([33, 70], 1) == [80]
    assert [geometry_rank_for(r, [10, 20]) for r in (2, 9, 10, 11, 20)] == [10, 10, 10, 20, 20]
    assert geometry_rank_for(21, [10, 20]) == 0 and geometry_rank_for(7, []) == 0
    # the same classes as a plan on the matrix gives (layout geometry of rank 3 under [3, 10, 20])
    M = C.CountMatrix(_matrix())
    h0, v0 = _layout_view(M, 0, 20)
    want = (v0.row_slots, v0.block_width)
    C.load().vbnmf_layout_destroy(h0)
    M.plan_ranks([3, 10, 20])
    h1, v1 = _layout_view(M, 0, 3)
    assert (v1.row_slots, v1.block_width) == want
    C.load().vbnmf_layout_destroy(h1)
    M.close()


def test_vb_factorize_leaves_a_callers_plan_alone():
    """vb_factorize used to set its own plan on the caller's CountMatrix and clear it afterwards."""
    import ccfindr_amd as C
    from fake_engine import NumpyPhaseEngine
    X = _matrix().toarray()[:60, :80]
    X = X[X.sum(1) > 0][:, X[X.sum(1) > 0].sum(0) > 0]
    M = C.CountMatrix(X)
    M.plan_ranks([2, 6])
    h, before = _layout_view(M, 1, 2)
    stride = before.row_slots
    C.load().vbnmf_layout_destroy(h)
    C.vb_factorize(M, ranks=[2, 3], nrun=1, verbose=0, Itmax=3, seed=5, engine_factory=lambda mm, rk: NumpyPhaseEngine(X, rk))
    h, after = _layout_view(M, 1, 2)
    assert after.row_slots == stride                                    # still the plan [2, 6]: stride of rank 6
    C.load().vbnmf_layout_destroy(h)
    M.close()


# ---- the stateless cache's key ------------------------------------------------------------
MASK = (1 << 64) - 1


def _chunk_digest(words, seed, c=0):
    """ccfindr_amd/csrc/engine.hip hash_bytes2, one chunk of whole 8-byte words: four chains, word q feeds chain q mod 4 (a
    last incomplete group of words feeds chains 0, 1, ... in order), the chains combined in order.  Returns the digest and,
    per word, the state of ITS chain before the word entered."""
    K, KL, KC = 0xFF51AFD7ED558CCD, 0xA24BAED4963EE407, 0xC4CEB9FE1A85EC53
    salt = 0x9E3779B97F4A7C15 ^ ((c * 0xD6E8FEB86659FD93) & MASK)
    h = [(seed ^ salt ^ ((j * KL) & MASK)) & MASK for j in range(4)]
    trace = []
    for q, w in enumerate(words):
        j = q % 4
        trace.append(h[j])
        h[j] = ((h[j] ^ w) * K) & MASK
        h[j] ^= h[j] >> 32
    d = h[0]
    for j in range(1, 4):
        d = ((d ^ h[j]) * KC) & MASK
        d ^= d >> 29
    return d, trace


def _whole_hash(words, seed):
    d, _ = _chunk_digest(words, seed)
    h = ((seed ^ d) * 0xC4CEB9FE1A85EC53) & MASK
    return h ^ (h >> 29)


def test_two_seeds_give_independent_hashes():
    """Two buffers built to collide in one chain under seed A (a later word of the same chain cancels the state difference an
    earlier word made) hash alike under A and differently under B: the key is 128 bits, not 64.  Also pins the function
    itself against the model above (the value must not depend on the library's thread count or build)."""
    from ccfindr_amd import _native as N
    L = N.load()
    A, B = 0x64656E7365, 0x3243F6A8885A308D
    rng = np.random.default_rng(5)
    w = [int(v) for v in rng.integers(0, 1 << 63, size=11, dtype=np.int64)]
    v = list(w)
    v[2] ^= 0x5DEECE66D                                                  # differ at word 2 (chain 2) ...
    _, tw = _chunk_digest(w[:7], A)
    _, tv = _chunk_digest(v[:7], A)
    v[6] = w[6] ^ tw[6] ^ tv[6]                                          # ... and cancel the state difference at word 6, same chain
    assert _chunk_digest(w, A)[0] == _chunk_digest(v, A)[0] and w != v

    def lib_hash(words, seed):
        buf = np.asarray(words, dtype=np.uint64)
        return int(L.vbnmf_test_hash_bytes(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes, seed))

    assert lib_hash(w, A) == _whole_hash(w, A) and lib_hash(w, B) == _whole_hash(w, B)     # the library computes the model
    assert lib_hash(w, A) == lib_hash(v, A)                              # the constructed collision is real ...
    assert lib_hash(w, B) != lib_hash(v, B)                              # ... and the second seed tells the buffers apart
    assert lib_hash(w, A) != lib_hash(w, B)


def test_hash_does_not_depend_on_the_thread_count():
    """Chunks (4 MB) are hashed by whatever threads there are and combined in order."""
    from ccfindr_amd import _native as N
    from ccfindr_amd.engine import host_threads, set_host_threads
    L = N.load()
    rng = np.random.default_rng(8)
    buf = rng.integers(0, 255, size=(9 << 20) + 13, dtype=np.uint8)      # three chunks, the last ragged, a byte tail
    got = []
    try:
        for nt in (1, 2, 7):
            set_host_threads(nt)
            assert host_threads() == nt
            got.append(int(L.vbnmf_test_hash_bytes(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes, 77)))
    finally:
        set_host_threads(0)
    assert got[0] == got[1] == got[2]
    buf[5 << 20] ^= 1
    assert int(L.vbnmf_test_hash_bytes(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes, 77)) != got[0]


@pytest.mark.parametrize("wide", [False, True])
def test_share_and_attach_a_layout_without_copies(wide, tmp_path):
    """vbnmf_matrix_share_layout cuts a layout with its entry stream INSIDE a file of the memory file system (complete when
    the name appears); vbnmf_matrix_attach_layout maps it in another handle and uses it in place.  The attached layout must
    be the layout a plain cut gives (same blob through the copying export), and survive the file's unlinking."""
    import ccfindr_amd as C
    from ccfindr_amd import node
    X = _matrix(seed=9, wide=wide)
    plain = C.CountMatrix(X)                                      # reference: cut in ordinary memory
    sharer = C.CountMatrix(X)
    shell = C.CountMatrix.shell(sharer.meta())
    for side in (1, 0):
        nb = plain.layout_blob_size(side, 8, 256)
        want = bytearray(nb)
        plain.export_layout(side, 8, 256, want)
        path = os.path.join(node.shm_dir(), node.fresh_name(f"share{side}"))
        sharer.share_layout(side, 8, 256, path)
        assert os.path.exists(path) and not os.path.exists(path + ".part") and os.path.getsize(path) == nb
        assert open(path, "rb").read() == bytes(want)             # the file IS the blob, byte for byte
        shell.attach_layout(path)
        os.unlink(path)                                           # the mappings keep the memory
        for handle in (sharer, shell):                            # both now serve the layout from the mapping
            back = bytearray(nb)
            assert handle.export_layout(side, 8, 256, back) == nb and back == want
        # a second share of the SAME geometry (cached by now) falls back to writing a copy
        path2 = os.path.join(node.shm_dir(), node.fresh_name(f"again{side}"))
        sharer.share_layout(side, 8, 256, path2)
        assert open(path2, "rb").read() == bytes(want)
        os.unlink(path2)
    plain.close(); sharer.close(); shell.close()


def test_layouts_do_not_depend_on_the_host_thread_count():
    """vbnmf_set_host_threads lifts the library's thread count for one process (the builder of a node's layouts takes its
    waiting peers' cores): cell order, row-major copy and both tiled layouts come out the same bytes whatever the count."""
    import ccfindr_amd as C
    from ccfindr_amd.engine import host_threads, set_host_threads
    X = _matrix(seed=9)
    default = host_threads()
    assert default >= 1
    blobs = {}
    try:
        for nt in (1, 3, 7, 0):
            before = set_host_threads(nt)
            assert before >= 1 and host_threads() == (nt if nt else default)
            os.environ["VBNMF_CELL_ORDER"] = "1"                     # the ordering too (off by default at this size)
            M = C.CountMatrix(X)
            out = []
            for side in (1, 0):
                nb = M.layout_blob_size(side, 10, 256)
                blob = bytearray(nb)
                M.export_layout(side, 10, 256, blob)
                out.append(bytes(blob))
            blobs[nt] = out
            M.close()
    finally:
        os.environ.pop("VBNMF_CELL_ORDER", None)
        set_host_threads(0)
    for nt in (3, 7, 0):
        assert blobs[nt][0] == blobs[1][0] and blobs[nt][1] == blobs[1][1], nt


def test_row_major_copy_by_ranges_of_the_output():
    """csrc/host.cpp transpose_compressed: from 2^18 stored entries on, several threads divide the OUTPUT into ranges of rows
    (bisection of every sorted column, then count / scan / scatter per range).  Held against the one-thread form (one range: a
    plain counting transposition) through both callers: the gene side's layout, cut from the row-major copy in the cells'
    renumbered order, must be the same bytes; and a matrix ingested by rows must come back as the matrix it was."""
    import scipy.sparse as sp
    import ccfindr_amd as C
    from ccfindr_amd.engine import set_host_threads
    rng = np.random.default_rng(12)
    n, m = 1500, 5200
    X = sp.random(n, m, density=0.11, format="csc", random_state=rng, data_rvs=lambda k: rng.integers(1, 9, k).astype(np.float64))
    keep_r, keep_c = np.ones(n), np.ones(m)
    keep_r[rng.integers(0, n, 40)] = 0.0                                 # some empty genes, some empty cells, ragged ranges
    keep_c[rng.integers(0, m, 60)] = 0.0
    X = sp.csc_matrix(sp.diags(keep_r) @ X @ sp.diags(keep_c))
    X.eliminate_zeros()
    assert X.nnz > 3 * (1 << 18)
    blobs = {}
    try:
        for nt in (1, 5):
            set_host_threads(nt)
            os.environ["VBNMF_CELL_ORDER"] = "1"                         # the copy is made in the cells' renumbered order
            M = C.CountMatrix(X)
            nb = M.layout_blob_size(0, 10, 256)
            blob = bytearray(nb)
            M.export_layout(0, 10, 256, blob)
            blobs[nt] = bytes(blob)
            M.close()
            R = C.CountMatrix(X.tocsr())                                 # by rows: transposed into the canonical columns
            back = R.to_scipy().tocsc()
            back.sort_indices()
            ref = X.copy()
            ref.sort_indices()
            assert back.shape == ref.shape and np.array_equal(back.indptr, ref.indptr)
            assert np.array_equal(back.indices, ref.indices) and np.array_equal(back.data, ref.data)
            R.close()
    finally:
        os.environ.pop("VBNMF_CELL_ORDER", None)
        set_host_threads(0)
    assert blobs[5] == blobs[1]


def test_usable_cores_follows_the_affinity_mask_and_the_cgroup_quota(monkeypatch, tmp_path):
    """The layout builder of a node takes its waiting peers' cores -- as many as this process may REALLY use: the affinity mask
    cut down to the cgroup's CPU quota (round 5: 128 visible CPUs under a quota of 16 made the builder's 128 threads cut the
    layouts in 0.39 s instead of 0.18)."""
    import builtins
    import os
    from ccfindr_amd import node
    n = node.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            f = tmp_path / "cpu.max"
            f.write_text("300000 100000\n")                # a quota of three cores
            return real_open(f, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    assert node.usable_cores() == min(3, len(os.sched_getaffinity(0)))
