#!/bin/bash
# profiles/collect.sh TAG -- run on the GPU box from the repo root (through gpurun):
#   1. bench.py (full line, CPU baseline included)              -> gpurun_out/TAG_bench.json
#   2. rocprofv3 --kernel-trace --stats over the same command    -> gpurun_out/TAG_stats/
#   3. separate --pmc passes (FETCH_SIZE ; WRITE_SIZE ; SQ ; L2) -> gpurun_out/TAG_pmc_*/
# then profiles/summarize_pmc.py folds 2+3 into profiles-ready files under gpurun_out/TAG_summary/.
set -e
TAG=${1:-r04}
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench done"
export TMPDIR=/tmp
export BENCH_NO_TRAFFIC=1 BENCH_NO_SWEEP=1 BENCH_NO_SMALL=1        # (no nested profiler runs; the profiled passes hold the headline's kernels only: the C4 rank sweep would mix 19 ranks' k_sweep into the means)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o stats -- python3 $REPO/bench.py --steps 200 --warmup 10 --no-cpu > $OUT/${TAG}_stats_bench.json 2> $OUT/${TAG}_stats.err
echo "stats done"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $pass --output-format csv -d $OUT/${TAG}_pmc_$name -o pmc -- python3 $REPO/bench.py --steps 24 --warmup 2 --no-cpu > /dev/null 2> $OUT/${TAG}_pmc_$name.err
  echo "pmc $name done"
done
cd $REPO
python3 profiles/summarize_pmc.py $TAG
