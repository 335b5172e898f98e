#!/bin/bash
# abrank.sh "ranks" libA libB ...: step / sweep time by rank on the C3 matrix, same box
ranks=$1; shift
mkdir -p gpurun_out
for r in $ranks; do
for lib in "$@"; do
  VBNMF_LIB=$PWD/gpurun_build/libs/$lib python bench.py --rank $r --steps 200 --warmup 10 --no-cpu --no-ml 2>/dev/null > gpurun_out/abr_$lib.$r.log || { echo "$lib rank $r failed"; continue; }
  python - <<EOF
import json; d=json.load(open("gpurun_out/abr_$lib.$r.log")); print("rank $r $lib value %.1f step %.4f ms sweep %.4f ms" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"]))
EOF
done
done
