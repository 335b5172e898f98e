#!/usr/bin/env python3
"""profiles/ubench/svd_copies.py -- the truncated SVD of the svd2 initialiser on a C3-shaped matrix, for
`rocprofv3 --memory-copy-trace --kernel-trace --stats`: the device-resident form (vbnmf_engine_svd) must show a copy
count that does not grow with the iteration count (ids, tables and the final download only), against the host-QR form,
which moves every operand and result of every sparse product through the host."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ccfindr_amd as C              # noqa: E402
from ccfindr_amd import synth        # noqa: E402
from ccfindr_amd.linalg import truncated_svd   # noqa: E402

method = sys.argv[1] if len(sys.argv) > 1 else "device"
n, m, k = 20000, 50000, 10
depth = np.round(np.random.default_rng(3).lognormal(np.log(1500.0), 0.3, size=m)).astype(np.int64)
X = synth.fill_empty(synth.simulate_data(n, [m // k] * k, alpha0=0.065, seed=3, depth=depth), seed=3)
M = C.CountMatrix(X)
t0 = time.perf_counter()
u, d, vt = truncated_svd(M, k, method=method, maxit=30)
dt = time.perf_counter() - t0
print(f"method {method}: {dt:.2f} s; leading singular values {np.round(d[:4], 3).tolist()}; "
      f"orthonormality {np.max(np.abs(u.T @ u - np.eye(k))):.2e} / {np.max(np.abs(vt @ vt.T - np.eye(k))):.2e}")
