#!/bin/bash
# pair_prof.sh: kernel durations (rocprofv3 --kernel-trace) of the device-driven loop with both updates in one launch
# (default) and with the two-launch form (VBNMF_NO_UPDATE_PAIR=1), ranks 10 and 20, same box.
export BENCH_NO_SWEEP=1 BENCH_NO_TRAFFIC=1 TMPDIR=/tmp
R=$PWD
cd /tmp
for rank in 10 20; do
for np in 0 1; do
  export VBNMF_NO_UPDATE_PAIR=$np VBNMF_UPDATE_PAIR=$((1-np))
  rm -rf /tmp/pp_$np
  rocprofv3 --kernel-trace --output-format csv -d /tmp/pp_$np -o t -- python3 $R/bench.py --rank $rank --steps 200 --warmup 5 --no-cpu --no-ml --no-traffic > /dev/null 2>&1
  echo "== rank $rank VBNMF_NO_UPDATE_PAIR=$np"; python3 $R/profiles/ubench/r05/kernel_means.py /tmp/pp_$np
done
done
