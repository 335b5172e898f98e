#!/bin/bash
# byrank.sh "ranks": step / sweep time by rank on the C3 matrix with the in-tree library (median of 5 x 200 steps)
mkdir -p gpurun_out
for r in $1; do
  python bench.py --rank $r --steps 200 --warmup 10 --no-cpu --no-ml 2>/dev/null > gpurun_out/byrank_$r.log || { echo "rank $r failed"; continue; }
  python - <<EOF
import json; d=json.load(open("gpurun_out/byrank_$r.log")); print("rank $r: %.1f it/s, step %.1f us, k_sweep %.1f us (HIP events), fp64 frac %.3f" % (d["value"], 1e3*d["ms_per_step"], 1e3*d["roofline"]["kernel_ms"], d["roofline_fp64"]["frac"]))
EOF
done
