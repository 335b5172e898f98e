#!/bin/bash
# profiles/collect_r04_evidence.sh TAG -- the evidence VERDICT r03 (Next #4) found missing, collected on the GPU box from
# the repo root through gpurun: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (each its own run, no trace
# domain beside --pmc) of (a) `bench.py --rank 20` on the C3 matrix and (b) one C5 partitioned run (8 partitions on the
# one GPU, tests/manual_c5_check.py).  profiles/summarize_kernels.py folds the raw CSVs into per-kernel tables.
set -e
TAG=${1:-r04}
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B20="$REPO/bench.py --rank 20 --no-cpu --no-ml"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_rank20_stats -o stats -- python3 $B20 --steps 200 --warmup 10 > $OUT/${TAG}_rank20_bench.json 2> $OUT/${TAG}_rank20_stats.err
echo "rank20 stats done"
for pass in FETCH_SIZE WRITE_SIZE; do
  name=$(echo $pass | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $pass --output-format csv -d $OUT/${TAG}_rank20_pmc_$name -o pmc -- python3 $B20 --steps 24 --warmup 2 > /dev/null 2> $OUT/${TAG}_rank20_pmc_$name.err
  echo "rank20 pmc $name done"
done
C5="$REPO/tests/manual_c5_check.py --no-oracle"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c5_stats -o stats -- python3 $C5 --timing-steps 50 > $OUT/${TAG}_c5_stats.log 2> $OUT/${TAG}_c5_stats.err
echo "c5 stats done"
for pass in FETCH_SIZE WRITE_SIZE; do
  name=$(echo $pass | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $pass --output-format csv -d $OUT/${TAG}_c5_pmc_$name -o pmc -- python3 $C5 --timing-steps 10 > /dev/null 2> $OUT/${TAG}_c5_pmc_$name.err
  echo "c5 pmc $name done"
done
cd $REPO
python3 profiles/summarize_kernels.py $TAG rank20 c5
