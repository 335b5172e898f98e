#!/bin/bash
mkdir -p gpurun_out/r3j
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3j/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3j/tests.log
timeout -k 10 300 python bench.py --steps 300 --warmup 10 --no-cpu --no-ml > gpurun_out/r3j/bench.json 2>/dev/null
echo "C3 $(grep -o '"value": [0-9.]*\|kernel_ms": [0-9.]*\|value_host_stepped": [0-9.]*' gpurun_out/r3j/bench.json | head -3 | tr '\n' ' ')"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3j/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu --no-ml > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3j/stats.err
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r3j/stats -name '*kernel_stats.csv' | head -1)
python3 - $f <<'PY'
import csv,sys
for i,row in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i<4: print(row["Name"][:50], row["Calls"], row["AverageNs"])
PY
