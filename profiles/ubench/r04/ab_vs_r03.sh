#!/bin/bash
# Round 3's final tree (git b89bd7d, built under build/r03_tree) against this round's build: same box, interleaved, the
# headline command (device-driven K = 1000; host-stepped; k_sweep by HIP events) and ranks 16 / 20; round 3 runs first.
export BENCH_NO_SWEEP=1 BENCH_NO_TRAFFIC=1
R=$PWD
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s (host-stepped %.1f) step %.1f us k_sweep %.1f us frac %.4f' % (d['value'], d.get('value_host_stepped', d['host_stepped']['value']), 1e3*d['ms_per_step'], 1e3*d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
for rep in 1 2 3; do
  for r in 10 16 20; do
    if [ $r = 10 ]; then a=""; else a="--rank $r"; fi
    echo "rank $r rep $rep r03: $(cd $R/build/r03_tree && python3 bench.py $a --no-cpu --no-ml 2>/dev/null | line)"
    echo "rank $r rep $rep r04: $(cd $R && python3 bench.py $a --no-cpu --no-ml 2>/dev/null | line)"
  done
done
