#!/bin/bash
# one C5 partition alone: kernel stats + FETCH/WRITE passes, cell order on (default) and off (run through gpurun)
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
cd /tmp
for ord in auto 0; do
  if [ $ord = auto ]; then unset VBNMF_CELL_ORDER; else export VBNMF_CELL_ORDER=$ord; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_c5part_${ord}_stats -o stats -- python3 $REPO/profiles/ubench/r04/c5_one_partition.py > $OUT/r04_c5part_${ord}.json 2> $OUT/r04_c5part_${ord}.err
  echo "order $ord: $(cat $OUT/r04_c5part_${ord}.json)"
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/r04_c5part_${ord}_stats/**/*kernel_stats.csv", recursive=True))[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.3:
        print("   %-70s calls %6s avg %9.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  find $OUT/r04_c5part_${ord}_stats -name "*kernel_trace.csv" -delete
done
unset VBNMF_CELL_ORDER
for pass in FETCH_SIZE WRITE_SIZE; do
  name=$(echo $pass | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $pass --output-format csv -d $OUT/r04_c5part_pmc_$name -o pmc -- python3 $REPO/profiles/ubench/r04/c5_one_partition.py --steps 20 > /dev/null 2> $OUT/r04_c5part_pmc_$name.err
done
cd $REPO
python3 profiles/summarize_kernels.py r04 c5part
