#!/usr/bin/env python3
"""profiles/ubench/r05/setup_times.py -- where the set-up seconds of the headline matrix go on this box: ingestion, the cell
order + row-major copy, the cut of each side (VBNMF_BUILD_TIMES=1 prints the phases), engine creation with the two sides cut
side by side (default) and one after the other (VBNMF_SERIAL_SIDES=1), a second engine on the cached layouts."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:                       # child: one measurement in a fresh process (module-level caches, env switches)
    import numpy as np, scipy.sparse as sp, ccfindr_amd as C
    z = np.load("/tmp/vbnmf_c3.npz")
    X = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
    from ccfindr_amd.engine import device_warmup
    device_warmup(0)
    t0 = time.perf_counter(); M = C.CountMatrix(X); t1 = time.perf_counter()
    e = C.VBEngine(M, 10); t2 = time.perf_counter()
    e2 = C.VBEngine(M, 10); t3 = time.perf_counter()
    print(f"{sys.argv[1]:28s} ingest {t1 - t0:.3f} s  engine_create {t2 - t1:.3f} s  second engine {t3 - t2:.4f} s", flush=True)
    sys.exit(0)
import numpy as np, bench
name, X, r = bench.make_workload(False)
X = X.tocsc()
np.savez("/tmp/vbnmf_c3.npz", data=X.data, indices=X.indices, indptr=X.indptr, shape=np.asarray(X.shape))
for rep in range(2):
    for tag, env in (("sides side by side", {}), ("sides one after the other", {"VBNMF_SERIAL_SIDES": "1"})):
        subprocess.run([sys.executable, os.path.abspath(__file__), tag], env=dict(os.environ, **env), check=True)
subprocess.run([sys.executable, os.path.abspath(__file__), "phases (serial)"], env=dict(os.environ, VBNMF_SERIAL_SIDES="1", VBNMF_BUILD_TIMES="1"), check=True)
subprocess.run([sys.executable, os.path.abspath(__file__), "phases (side by side)"], env=dict(os.environ, VBNMF_BUILD_TIMES="1"), check=True)
