#!/bin/bash
mkdir -p gpurun_out/r3i
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3i/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3i/tests.log
for ub in default 256; do
  if [ $ub = default ]; then unset VBNMF_UPDATE_BLOCKS; else export VBNMF_UPDATE_BLOCKS=$ub; fi
  echo "UPDATE_BLOCKS=$ub"
  timeout -k 10 120 python gpurun_build/small_len.py c1 2>/dev/null
  timeout -k 10 120 python gpurun_build/small_len.py c2 2>/dev/null
done
unset VBNMF_UPDATE_BLOCKS
timeout -k 10 300 python bench.py --steps 300 --warmup 10 --no-cpu --no-ml > gpurun_out/r3i/bench.json 2>/dev/null
echo "C3 $(grep -o '"value": [0-9.]*\|kernel_ms": [0-9.]*\|value_host_stepped": [0-9.]*' gpurun_out/r3i/bench.json | head -3 | tr '\n' ' ')"
