#!/bin/bash
export BENCH_NO_TRAFFIC=1      # no nested rocprofv3 runs under a profiler (bench.py: measure_sweep_traffic)
# per-side k_update times at ranks 10 and 20 (run through gpurun from the repo root)
export TMPDIR=/tmp BENCH_NO_SWEEP=1
REPO=$PWD
cd /tmp
for r in 10 20; do
  for ord in 1 0; do
    rm -rf /tmp/us_$r_$ord
    VBNMF_CELL_ORDER=$ord rocprofv3 --kernel-trace --output-format csv -d /tmp/us_${r}_${ord} -o t -- python3 $REPO/bench.py --rank $r --steps 100 --warmup 5 --no-cpu --no-ml > /dev/null 2>&1
    echo "rank $r cell order $ord"
    python3 $REPO/profiles/ubench/r04/update_split.py /tmp/us_${r}_${ord}
  done
done
