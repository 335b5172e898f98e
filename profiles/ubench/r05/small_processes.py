"""The same workload as small_concurrent.py, but P PROCESSES side by side on the one GPU (each with its own HIP runtime) instead of
K host threads in one: tells a limit of the runtime inside a process from a limit of the device."""
import os, sys, time, subprocess
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 2:                      # child: wait for the go file, run, report
    import ccfindr_amd as C
    from ccfindr_amd import synth
    X = synth.drop_empty(synth.simulate_data(1030, (150, 150, 150), seed=3, sparse=True))
    M = C.CountMatrix(X)
    ranks = list(range(2, 10)); nrun = int(sys.argv[2]); Itmax = 600
    C.vb_factorize(M, ranks=[2, 3], nrun=1, verbose=0, Tol=0.0, seed=1, Itmax=20)
    print("ready", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    C.vb_factorize(M, ranks=ranks, nrun=nrun, verbose=0, Tol=0.0, seed=5, Itmax=Itmax, unif_stop=False, hyper_update_n0=10)
    print(f"done {time.perf_counter() - t0:.4f} {len(ranks) * nrun * Itmax}", flush=True)
    sys.exit(0)
for P in (1, 2, 4):
    nrun = 8 // P                           # the same total work whatever P
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(nrun)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
             for _ in range(P)]
    for p in procs:
        while "ready" not in p.stdout.readline():
            pass
    t0 = time.perf_counter()
    for p in procs:
        p.stdin.write("go\n"); p.stdin.flush()
    its = 0
    for p in procs:
        line = p.stdout.readline().split()
        its += int(line[2])
        p.wait()
    dt = time.perf_counter() - t0
    print(f"   {P} process(es): {dt:6.3f} s  {its / dt:9.0f} iterations/s in all", flush=True)
