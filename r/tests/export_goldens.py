#!/usr/bin/env python3
"""r/tests/export_goldens.py [outdir] -- the committed single-step fixtures tests/golden/step_*.npz and the reference's bundled
PBMC sample (tests/golden/pbmc_extdata_r5.npz) as plain little-endian float64 files R can read with readBin (R has no npz
reader): <outdir>/<case>/<name>.f64 + <outdir>/<case>/dims.txt (one line per array: name nrow ncol).  Needs numpy only.
r/tests/parity.R reads them."""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "golden_bin")
files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "step_*.npz"))) + [os.path.join(ROOT, "tests", "golden", "pbmc_extdata_r5.npz")]
for f in files:
    d = dict(np.load(f))
    case = os.path.splitext(os.path.basename(f))[0]
    if "X" not in d:                                         # the PBMC sample is stored as compressed columns: densify for R's matrix()
        n, m = int(d.pop("n")), int(d.pop("m"))
        indptr, indices, data = d.pop("indptr"), d.pop("indices"), d.pop("data")
        X = np.zeros((n, m))
        for j in range(m):
            X[indices[indptr[j]:indptr[j + 1]], j] = data[indptr[j]:indptr[j + 1]]
        d["X"] = X
        d.setdefault("hyper", np.ones(4))                    # (tests/golden/make_golden.py: aw = bw = ah = bh = 1, fudge = eps)
        d.setdefault("fudge", np.finfo(np.float64).eps)
    os.makedirs(os.path.join(out, case), exist_ok=True)
    with open(os.path.join(out, case, "dims.txt"), "w") as fh:
        for k in d:
            a = np.asarray(d[k], dtype=np.float64)
            a2 = a.reshape(-1, 1) if a.ndim < 2 else a
            np.asfortranarray(a2).ravel(order="F").astype("<f8").tofile(os.path.join(out, case, k + ".f64"))
            fh.write(f"{k} {a2.shape[0]} {a2.shape[1]}\n")
    print(case, {k: tuple(np.asarray(d[k]).shape) for k in d})
