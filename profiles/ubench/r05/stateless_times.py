"""The literal drop-in call (vbnmf_update_csc / _dense behind .Call("_ccfindR_vbnmf_update")), repeated
on the same matrix: seconds per call and what they are made of -- the hash of X that keys the stateless cache, the state in,
the priming sweep + the step, the state out.  Usage: stateless_times.py [small]   (VBNMF_LIB picks a library variant)."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
import ccfindr_amd as C  # noqa: E402
from ccfindr_amd import _native as N, synth  # noqa: E402
from ccfindr_amd.bayesian import vbnmf_update  # noqa: E402

small = len(sys.argv) > 1 and sys.argv[1] == "small"
name, X, r = bench.make_workload(small)
X = X.tocsc()
X.sort_indices()
n, m = X.shape
hy = {"aw": 0.1, "bw": 1.0, "ah": 0.1, "bh": 1.0}
wh = synth.random_state(n, m, r, hy, seed=1)
L = N.load()


def med(f, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), float(min(ts))


t0 = time.perf_counter()
out = vbnmf_update(X, wh, hy)
first = time.perf_counter() - t0
t_call = med(lambda: vbnmf_update(X, wh, hy), 9)
print(f"{name}: nnz {X.nnz}, first call {first:.3f} s, repeat call median {1e3 * t_call[0]:.2f} ms (best {1e3 * t_call[1]:.2f})")

ind = np.ascontiguousarray(X.indices, dtype=np.int32)
val = np.ascontiguousarray(X.data, dtype=np.float64)


def hash_once():
    L.vbnmf_test_hash_bytes(ind.ctypes.data_as(ctypes.c_void_p), ind.nbytes, 1)
    L.vbnmf_test_hash_bytes(val.ctypes.data_as(ctypes.c_void_p), val.nbytes, 2)


t_hash = med(hash_once, 9)
print(f"   one pass over the index and value arrays ({(ind.nbytes + val.nbytes) / 1e6:.0f} MB) by vbnmf_test_hash_bytes: {1e3 * t_hash[0]:.2f} ms")

Mx = C.CountMatrix(X)
eng = C.VBEngine(Mx, r)
t_in = med(lambda: eng.set_state(wh["lw"], wh["lh"], wh["eh"]), 9)
t_step = med(lambda: eng.step(hy), 9)
t_out = med(lambda: eng.get_state(), 9)
print(f"   resident engine: set_state (conversion, copies in, priming sweep) {1e3 * t_in[0]:.2f} ms, step {1e3 * t_step[0]:.3f} ms, "
      f"get_state (six arrays out, conversion) {1e3 * t_out[0]:.2f} ms")
eng.close()
Mx.close()

if small:
    A = np.asfortranarray(X.toarray())
    t0 = time.perf_counter()
    vbnmf_update(A, wh, hy)
    first = time.perf_counter() - t0
    t_d = med(lambda: vbnmf_update(A, wh, hy), 9)
    print(f"   dense form of the same matrix ({A.nbytes / 1e6:.0f} MB): first call {first:.3f} s, repeat {1e3 * t_d[0]:.2f} ms")
