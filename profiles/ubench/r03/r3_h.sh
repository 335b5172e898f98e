#!/bin/bash
export TMPDIR=/tmp
cd /tmp
for w in c1 c2; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3h/$w -o stats -- python3 $GRAFT_REPO_ROOT/gpurun_build/small_len.py $w > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3h_$w.err
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r3h/$w -name '*kernel_stats.csv' | head -1)
  echo "== $w"; python3 - $f <<'PY'
import csv,sys
for i,row in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i<6: print(row["Name"][:60], row["Calls"], row["AverageNs"], row["MinNs"])
PY
done
