#!/bin/bash
# abenv.sh "ENV1=.. " "ENV2=..": same library, different environment settings, same box; two rounds
mkdir -p gpurun_out
i=0
for rep in 1 2; do
for envs in "$@"; do
  i=$((i+1))
  env $envs python bench.py --steps 400 --warmup 10 --no-cpu --no-ml 2>/dev/null > gpurun_out/abenv_$i.log || { echo "[$envs] failed"; continue; }
  python - <<EOF
import json; d=json.load(open("gpurun_out/abenv_$i.log")); print("[$envs] rep$rep value %.1f step %.4f ms host %.1f sweep %.4f ms" % (d["value"], d["ms_per_step"], d["host_stepped"]["value"], d["roofline"]["kernel_ms"]))
EOF
done
done
