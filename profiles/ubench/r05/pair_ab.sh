#!/bin/bash
# pair_ab.sh: both posterior updates in one launch (k_update2, default) against the two-launch form (VBNMF_NO_UPDATE_PAIR=1),
# same library, same box, interleaved; ranks 10 and 20 of the C3 matrix.  BENCH_NO_TRAFFIC=1: no nested rocprofv3 runs.
export BENCH_NO_TRAFFIC=1
mkdir -p gpurun_out
for rank in 10 20; do
for rep in 1 2 3; do
for envs in "VBNMF_NO_UPDATE_PAIR=1" "VBNMF_UPDATE_PAIR=1"; do
  env $envs python bench.py --steps 400 --warmup 10 --no-cpu --no-ml --no-traffic --rank $rank 2>/dev/null > gpurun_out/pair_ab.log || { echo "[$envs] failed"; continue; }
  python - <<PY
import json; d=json.load(open("gpurun_out/pair_ab.log")); print("rank $rank [$envs] rep$rep value %.1f step %.4f ms host %.1f sweep %.4f ms" % (d["value"], d["ms_per_step"], d["host_stepped"]["value"], d["roofline"]["kernel_ms"]))
PY
done
done
done
