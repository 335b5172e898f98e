#!/usr/bin/env python3
"""profiles/ubench/r05/small_lib_ab.py LIB... -- C1, the PBMC-sized sample, C2 and a mid-size sparse matrix: microseconds per step of
the device-driven loop under each library variant (profiles/ubench/libs/LIB through VBNMF_LIB, a child process per measurement), same box,
interleaved.  An argument NAME=VALUE sets an environment variable for the in-tree library instead."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
HY = {"aw": 1.0, "bw": 1.0, "ah": 1.0, "bh": 1.0}
CASES = ["C1", "PBMC", "C2", "MID"]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, ccfindr_amd as C
    from ccfindr_amd import synth
    case = sys.argv[2]
    X, r = {"C1": lambda: (synth.drop_empty(synth.simulate_data(200, (100, 150, 250), seed=1, sparse=False)), 3),
            "PBMC": lambda: (synth.fill_empty(synth.simulate_data(1030, [150] * 3, alpha0=0.3, seed=4, depth=np.full(450, 900))), 5),
            "C2": lambda: (synth.fill_empty(synth.simulate_data(2000, [2000] * 5, alpha0=2.0, seed=2, depth=np.full(10000, 4000)), seed=2), 5),
            "MID": lambda: (synth.fill_empty(synth.simulate_data(5000, [4000] * 5, alpha0=0.1, seed=3, depth=np.full(20000, 400))), 8)}[case]()
    n, m = X.shape
    eng = C.VBEngine(C.CountMatrix(X), r)
    wh = synth.random_state(n, m, r, HY, seed=1000 + r)
    eng.set_state(wh["lw"], wh["lh"], wh["eh"])
    eng.run(HY, Itmax=300, Tol=0.0, flags=(False,) * 4)
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        res = eng.run(HY, Itmax=3000, Tol=0.0, n0=10, dn=1, flags=(False,) * 4)
        best = max(best, res["it"] / (time.perf_counter() - t0))
    print(f"{1e6 / best:.2f}")
    sys.exit(0)
libs = sys.argv[1:]
for case in CASES:
    for rep in range(2):
        for lib in libs:
            env = dict(os.environ)
            if "=" in lib:                                 # an environment setting instead of a library variant
                k, v = lib.split("=", 1); env[k] = v
            elif lib != "tree":
                env["VBNMF_LIB"] = os.path.join(ROOT, "profiles", "ubench", "libs", lib)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", case], env=env, capture_output=True, text=True)
            print(f"{case:5s} rep {rep} [{lib}] {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else 'failed: ' + out.stderr[-300:]} us per step", flush=True)
