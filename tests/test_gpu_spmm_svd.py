"""GPU: the sparse products of the tiled layout (vbnmf_engine_spmm / k_spmm) and the truncated SVD built on them
(ccfindr_amd/linalg.py), which stands in for irlba in the svd2 initialiser (reference R/bayesian.R:150-159).
Checker: dense numpy products and numpy's full SVD."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def counts(n, m, lam, seed):
    rng = np.random.default_rng(seed)
    X = rng.poisson(lam, size=(n, m)).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1
    X[rng.integers(0, n, m), np.arange(m)] += 1
    return X


@pytest.mark.parametrize("n,m,r,lam", [(64, 96, 4, 1.5), (300, 400, 10, 0.05), (130, 70, 1, 1.0), (97, 211, 7, 0.3),
                                       (50, 60, 20, 2.0), (40, 45, 32, 1.0), (2500, 4300, 12, 0.1)])
def test_spmm_both_orientations(n, m, r, lam):
    import ccfindr_amd as C
    X = counts(n, m, lam, seed=n + r)
    rng = np.random.default_rng(1)
    eng = C.VBEngine(C.CountMatrix(X), r)
    B = rng.standard_normal((r, m))
    got = eng.spmm(B)
    want = X @ B.T
    assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))
    W = rng.standard_normal((n, r))
    got = eng.spmm(W, transpose=True)
    want = W.T @ X
    assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want))
    # the engine's state is gone after a product
    with pytest.raises(C.VBNMFError):
        eng.step({"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0})
    eng.close()


def test_spmm_wide_values_and_sparse_input():
    import ccfindr_amd as C
    X = counts(80, 150, 0.7, seed=5)
    X = X * (np.median(X.sum(axis=0)) / X.sum(axis=0))[None, :]
    X[3, 4] = 70000.25
    rng = np.random.default_rng(2)
    eng = C.VBEngine(C.CountMatrix(sp.csr_matrix(X)), 5)
    B = rng.standard_normal((5, 150))
    want = X @ B.T
    assert np.max(np.abs(eng.spmm(B) - want)) <= 1e-12 * np.max(np.abs(want))
    eng.close()


def planted(n, m, k, seed):
    """Counts with k well separated components, so the spectrum has a gap after k."""
    rng = np.random.default_rng(seed)
    W = rng.gamma(0.3, 1.0, size=(n, k))
    H = np.zeros((k, m))
    H[rng.integers(0, k, m), np.arange(m)] = rng.uniform(2, 6, m)
    X = rng.poisson(W @ H).astype(np.float64)
    X[np.arange(n), rng.integers(0, m, n)] += 1
    X[rng.integers(0, n, m), np.arange(m)] += 1
    return X


def test_svd_initialiser_on_device_matches_host_svd_up_to_the_triplets_signs():
    """vb_init(initializer = 'svd') (reference R/bayesian.R:116-149) reads the leading `rank` triplets only: beyond
    min(nrow, ncol)/2 > rank they come from the device's truncated SVD.  The formulas depend on each triplet's sign
    (LAPACK's in the reference, arbitrary): the host's triplets are aligned to the device's before comparing."""
    from ccfindr_amd import bayesian, linalg
    import ccfindr_amd as C
    from oracle import vbnmf_oracle as O
    X = planted(300, 500, 4, seed=11)
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
    rank = 4
    got = bayesian.vb_init(300, 500, sp.csc_matrix(X), rank, hy, "svd", rng=np.random.default_rng(3))
    assert (got["w"] >= 0).all() and (got["h"] >= 0).all() and not got["dw"].any()
    U, D, Vt = np.linalg.svd(X, full_matrices=False)
    # the leading pair is sign-free: the Perron pair
    assert np.allclose(np.outer(got["w"][:, 0], got["h"][0]), D[0] * np.outer(U[:, 0], Vt[0]), rtol=1e-6, atol=1e-8)
    for k in range(1, rank):
        cands = []
        for sgn in (1.0, -1.0):                                    # the two sign choices of triplet k
            x, y = sgn * U[:, k], sgn * Vt[k]
            xp, yp = np.where(x > 0, x, 0.0), np.where(y > 0, y, 0.0)
            sig = np.linalg.norm(xp) * np.linalg.norm(yp)          # :132-138 (the positive branch is always taken)
            cands.append((np.sqrt(D[k] * sig) * xp / np.linalg.norm(xp), np.sqrt(D[k] * sig) * yp / np.linalg.norm(yp)))
        err = [max(np.max(np.abs(got["w"][:, k] - w)), np.max(np.abs(got["h"][k] - h))) for w, h in cands]
        assert min(err) <= 1e-6 * max(np.max(cands[0][0]), np.max(cands[0][1])), (k, err)
    res = C.vb_factorize(sp.csc_matrix(X), ranks=rank, nrun=1, initializer="svd", verbose=0, Itmax=30)
    assert res.basis[0].shape == (300, rank) and np.isfinite(res.measure["lml"][0])


@pytest.mark.parametrize("n,m,k", [(400, 900, 4), (1200, 700, 6)])
def test_truncated_svd_matches_full_svd(n, m, k):
    from ccfindr_amd.linalg import truncated_svd
    X = planted(n, m, k, seed=n)
    u, d, vt = truncated_svd(sp.csc_matrix(X), k)
    U, D, Vt = np.linalg.svd(X, full_matrices=False)
    assert np.max(np.abs(d / D[:k] - 1)) <= 1e-9
    for i in range(k):                                    # singular vectors up to sign
        assert abs(abs(u[:, i] @ U[:, i]) - 1) <= 1e-8 and abs(abs(vt[i] @ Vt[i]) - 1) <= 1e-8
    assert np.allclose(u.T @ u, np.eye(k), atol=1e-12) and np.allclose(vt @ vt.T, np.eye(k), atol=1e-12)


def test_svd2_initialiser_on_device_matches_host_svd():
    """vb_init(initializer = 'svd2') beyond min(nrow, ncol)/2 > rank: the reference calls irlba (R/bayesian.R:154)."""
    import ccfindr_amd as C
    from ccfindr_amd import bayesian
    X = planted(300, 500, 3, seed=7)
    hy = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 0.7}
    wh = bayesian.vb_init(300, 500, sp.csc_matrix(X), 3, hy, "svd2")
    U, D, Vt = np.linalg.svd(X, full_matrices=False)
    w, h = np.abs(U[:, :3]), np.abs(np.diag(D[:3]) @ Vt[:3])
    scale = hy["bh"] / np.mean(h)
    # subspace iteration leaves the vectors at ~1e-8 absolute: entries near zero cannot be held to a relative tolerance
    assert np.max(np.abs(wh["lw"] - w / scale)) <= 1e-6 * np.max(w / scale)
    assert np.max(np.abs(wh["lh"] - h * scale)) <= 1e-6 * np.max(h * scale)
    assert np.mean(wh["lh"]) == pytest.approx(hy["bh"])
    # and through the driver: one deterministic run from that start
    res = C.vb_factorize(sp.csc_matrix(X), ranks=3, nrun=1, initializer="svd2", verbose=0, Itmax=30)
    assert res.basis[0].shape == (300, 3) and np.isfinite(res.measure["lml"][0])


@pytest.mark.parametrize("n,m,k", [(400, 900, 4), (1200, 700, 6)])
def test_device_resident_svd_matches_full_svd_and_the_host_qr_form(n, m, k):
    """vbnmf_engine_svd: the whole subspace iteration on the device (CholeskyQR2 + Jacobi on the Gram matrix), against
    numpy's full SVD and against the host-QR form of the same iteration."""
    from ccfindr_amd.linalg import truncated_svd
    X = planted(n, m, k, seed=n)
    Xs = sp.csc_matrix(X)
    u, d, vt = truncated_svd(Xs, k, method="device")
    U, D, Vt = np.linalg.svd(X, full_matrices=False)
    assert np.max(np.abs(d / D[:k] - 1)) <= 1e-9
    for i in range(k):
        assert abs(abs(u[:, i] @ U[:, i]) - 1) <= 1e-8 and abs(abs(vt[i] @ Vt[i]) - 1) <= 1e-8
    assert np.allclose(u.T @ u, np.eye(k), atol=1e-10) and np.allclose(vt @ vt.T, np.eye(k), atol=1e-10)
    u2, d2, vt2 = truncated_svd(Xs, k, method="host_qr")
    assert np.max(np.abs(d / d2 - 1)) <= 1e-9


def test_device_svd_reports_a_rank_deficient_subspace():
    """Fewer independent directions than subspace columns: the Cholesky factor of the Gram matrix does not exist; the
    C entry says so (VBNMF_ERR_STATE) and truncated_svd continues in the QR form."""
    import ccfindr_amd as C
    from ccfindr_amd.linalg import truncated_svd
    rng = np.random.default_rng(3)
    A = rng.poisson(2.0, size=(60, 3)).astype(np.float64) + 1
    B = rng.poisson(2.0, size=(3, 80)).astype(np.float64) + 1
    X = A @ B                                              # rank 3 exactly
    eng = C.VBEngine(C.CountMatrix(X), 12)
    with pytest.raises(C.VBNMFError) as ei:
        eng.svd(2)
    assert ei.value.code == 5
    eng.close()
    u, d, vt = truncated_svd(sp.csc_matrix(X), 2)
    D = np.linalg.svd(X, compute_uv=False)
    assert np.max(np.abs(d / D[:2] - 1)) <= 1e-9


@pytest.mark.parametrize("k_engine,rank_out", [(40, 6), (48, 38), (64, 30)])
def test_device_svd_above_32_subspace_columns_with_no_fallback(k_engine, rank_out):
    """vbnmf_engine_svd called DIRECTLY with a subspace of 40 / 48 / 64 columns (the svd / svd2 initialisers at rank >= 23
    use k = rank + 10): the Gram, Cholesky, Jacobi and write-out kernels tile their 32 x 32 thread grid over the k x k
    matrices (csrc/init.h).  No host-QR fallback can satisfy this test: the triplets come from eng.svd() itself."""
    import ccfindr_amd as C
    n, m = 700, 900
    X = planted(n, m, 12, seed=5)                      # 12 strong components over full-rank Poisson noise
    eng = C.VBEngine(C.CountMatrix(sp.csc_matrix(X)), k_engine)
    u, d, vt, its = eng.svd(rank_out, tol=1e-7, maxit=100, seed=1)
    eng.close()
    U, D, Vt = np.linalg.svd(X, full_matrices=False)
    lead = 6                                            # a numpy model of the same iteration: values 2e-14, residuals 1e-7
    assert np.max(np.abs(d[:lead] / D[:lead] - 1)) <= 1e-9, (d[:lead], D[:lead])
    assert np.all(np.diff(d) <= 1e-12 * d[0])           # descending
    assert np.allclose(u.T @ u, np.eye(rank_out), atol=1e-9) and np.allclose(vt @ vt.T, np.eye(rank_out), atol=1e-9)
    # each returned triplet is a singular triplet of X to the iteration's accuracy: X v = d u
    resid = np.linalg.norm(X @ vt.T - u * d, axis=0) / D[0]
    assert np.max(resid[:lead]) <= 1e-5, resid[:lead]
