#!/bin/bash
# profiles/collect_r05_final.sh PART -- the round's closing collection on ONE box (run through gpurun from the repo root).
#   PART 1: bench line (default K and the driver's K = 20), rocprofv3 stats + PMC passes of the headline (collect.sh),
#           the sweep's per-workgroup / per-wave time line, the small configurations
#   PART 2: rank-20 evidence (kernel stats + traffic), C5 check, by-rank table
#   PART 3: the C4 rehearsal through the sharded driver
set -e
PART=${1:-1}
OUT=gpurun_out
mkdir -p $OUT
if [ "$PART" = 1 ]; then
  bash profiles/collect.sh r05 > $OUT/r05_collect.log 2>&1
  tail -2 $OUT/r05_collect.log
  python3 bench.py --steps 20 --warmup 5 > $OUT/r05_bench_k20.json 2> /dev/null
  python3 profiles/ubench/dbg_wg.py > $OUT/r05_sweep_workgroup_times.txt 2>&1 || true
  python3 tests/manual_config_table.py > $OUT/r05_configs.log 2>&1 || true
  cp $OUT/configs.json $OUT/r05_configs.json 2>/dev/null || true
elif [ "$PART" = 2 ]; then
  export TMPDIR=/tmp
  R=$PWD
  B20="$R/bench.py --rank 20 --no-cpu --no-ml"
  (cd /tmp && BENCH_NO_SWEEP=1 BENCH_NO_TRAFFIC=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/r05_rank20_stats -o stats -- python3 $B20 --steps 200 --warmup 10 > $R/$OUT/r05_rank20_bench.json 2> $R/$OUT/r05_rank20_stats.err)
  for pass in FETCH_SIZE WRITE_SIZE; do
    name=$(echo $pass | tr 'A-Z' 'a-z')
    (cd /tmp && BENCH_NO_SWEEP=1 BENCH_NO_TRAFFIC=1 rocprofv3 --pmc $pass --output-format csv -d $R/$OUT/r05_rank20_pmc_$name -o pmc -- python3 $B20 --steps 24 --warmup 2 > /dev/null 2> $R/$OUT/r05_rank20_pmc_$name.err)
  done
  python3 profiles/summarize_kernels.py r05 rank20 || true
  python3 tests/manual_c5_check.py > $OUT/r05_c5_check.log 2>&1 || true
  cp $OUT/c5_check.json $OUT/r05_c5_check.json 2>/dev/null || true
  BENCH_NO_SWEEP=1 bash profiles/ubench/byrank.sh "2 5 10 12 14 16 20 24 28 32" > $OUT/r05_by_rank.txt 2>&1
  cat $OUT/r05_by_rank.txt
else
  python3 tests/manual_c4_sharded.py --procs 1,2,4,6 > $OUT/r05_c4_sharded.log 2>&1
  cp $OUT/c4_sharded.json $OUT/r05_c4_sharded.json
  tail -5 $OUT/r05_c4_sharded.log
fi
