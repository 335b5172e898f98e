#!/bin/bash
# profiles/collect_r04_final.sh PART -- the round's closing collection on ONE box (run through gpurun from the repo root).
#   PART 1: bench line (K = 1000 and the driver's K = 20), rocprofv3 stats + PMC passes of the headline (collect.sh)
#   PART 2: rank-20 and C5 evidence after the round's changes, one C5 partition alone, C5 check, C4 rehearsal, by-rank table
set -e
PART=${1:-1}
OUT=gpurun_out
mkdir -p $OUT
if [ "$PART" = 1 ]; then
  bash profiles/collect.sh r04 > $OUT/r04_collect.log 2>&1
  tail -2 $OUT/r04_collect.log
  python3 bench.py --steps 20 --warmup 5 > $OUT/r04_bench_k20.json 2> /dev/null
  python3 tests/manual_config_table.py > $OUT/r04_configs.log 2>&1 || true
  cp $OUT/configs.json $OUT/r04_configs.json 2>/dev/null || true
else
  bash profiles/collect_r04_evidence.sh r04 > $OUT/r04_evidence.log 2>&1
  tail -2 $OUT/r04_evidence.log
  bash profiles/ubench/r04/c5_one_partition.sh > $OUT/r04_c5part.txt 2>&1
  python3 tests/manual_c5_check.py > $OUT/r04_c5_check.log 2>&1
  cp $OUT/c5_check.json $OUT/r04_c5_check.json
  python3 tests/manual_c4_sharded.py --procs 1,2,4,6 > $OUT/r04_c4_sharded.log 2>&1
  cp $OUT/c4_sharded.json $OUT/r04_c4_sharded.json
  BENCH_NO_SWEEP=1 bash profiles/ubench/byrank.sh "2 5 10 12 14 16 20 24 28 32" > $OUT/r04_by_rank.txt 2>&1
  cat $OUT/r04_by_rank.txt
fi
