#!/bin/bash
mkdir -p gpurun_out/r3d
for mode in 3 1 2; do
  VBNMF_FUSE_MODE=$mode timeout -k 10 300 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_device_loop.py tests/test_gpu_golden.py tests/test_gpu_parity.py -x -q > gpurun_out/r3d/tests_mode$mode.log 2>&1; echo "mode $mode tests rc=$?"; tail -3 gpurun_out/r3d/tests_mode$mode.log
done
for off in 0 1; do
  VBNMF_NO_FUSED_REDUCE=$off timeout -k 10 300 python bench.py --steps 300 --warmup 10 --no-cpu --no-ml > gpurun_out/r3d/ab_off$off.json 2>gpurun_out/r3d/ab_off$off.err || { echo "bench off=$off failed"; tail -3 gpurun_out/r3d/ab_off$off.err; }
  echo "NO_FUSED=$off $(grep -o '"value": [0-9.]*\|kernel_ms": [0-9.]*\|value_host_stepped": [0-9.]*' gpurun_out/r3d/ab_off$off.json | head -3 | tr '\n' ' ')"
done
for mode in 1 2; do
  VBNMF_FUSE_MODE=$mode timeout -k 10 300 python bench.py --steps 300 --warmup 10 --no-cpu --no-ml > gpurun_out/r3d/ab_mode$mode.json 2>/dev/null
  echo "MODE=$mode $(grep -o '"value": [0-9.]*\|kernel_ms": [0-9.]*\|value_host_stepped": [0-9.]*' gpurun_out/r3d/ab_mode$mode.json | head -3 | tr '\n' ' ')"
done
export TMPDIR=/tmp
cd /tmp
for off in 0; do
  VBNMF_NO_FUSED_REDUCE=$off timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3d/stats_off$off -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu --no-ml > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3d/stats_off$off.err
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r3d/stats_off$off -name '*kernel_stats.csv' | head -1)
  echo "== NO_FUSED=$off"; python3 - $f <<'PY'
import csv,sys
for i,row in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i<5: print(row["Name"][:50], row["Calls"], row["AverageNs"])
PY
done
