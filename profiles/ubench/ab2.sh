#!/bin/bash
# same-box A/B: ab2.sh libA libB ... -> value, step ms, sweep ms for each, two rounds; then kernel stats of each
mkdir -p gpurun_out
for rep in 1 2; do
for lib in "$@"; do
  VBNMF_LIB=$PWD/gpurun_build/libs/$lib python bench.py --steps 400 --warmup 10 --no-cpu --no-ml 2>/dev/null > gpurun_out/ab_$lib.$rep.log || { echo "$lib failed"; continue; }
  python - <<EOF
import json; d=json.load(open("gpurun_out/ab_$lib.$rep.log")); print("$lib rep$rep value %.1f step %.4f ms host %.1f sweep %.4f ms" % (d["value"], d["ms_per_step"], d["host_stepped"]["value"], d["roofline"]["kernel_ms"]))
EOF
done
done
