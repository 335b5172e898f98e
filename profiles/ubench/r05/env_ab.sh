#!/bin/bash
# env_ab.sh RANK "ENV1=.." "ENV2=..": same library, different environment settings, same box, interleaved, three rounds
export BENCH_NO_TRAFFIC=1 BENCH_NO_SWEEP=1
rank=$1; shift
for rep in 1 2 3; do
for envs in "$@"; do
  env $envs python3 bench.py --steps 400 --warmup 10 --no-cpu --no-ml --no-traffic --rank $rank 2>/dev/null > gpurun_out/env_ab.log || { echo "[$envs] failed"; continue; }
  python3 - <<PY
import json; d=json.load(open("gpurun_out/env_ab.log")); print("rank $rank [$envs] rep$rep value %.1f step %.4f ms host %.1f sweep %.4f ms" % (d["value"], d["ms_per_step"], d["host_stepped"]["value"], d["roofline"]["kernel_ms"]))
PY
done
done
