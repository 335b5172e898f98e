#!/bin/bash
mkdir -p gpurun_out/r3e
for rep in 1 2; do
for dbg in off 0 2 4 6; do
  if [ $dbg = off ]; then export VBNMF_NO_FUSED_REDUCE=1; unset VBNMF_FUSE_DEBUG; else unset VBNMF_NO_FUSED_REDUCE; export VBNMF_FUSE_DEBUG=$dbg; fi
  VBNMF_FUSE_MODE=1 timeout -k 10 300 python bench.py --steps 300 --warmup 10 --no-cpu --no-ml > gpurun_out/r3e/ab_$dbg.$rep.json 2>/dev/null
  echo "DBG=$dbg rep$rep $(grep -o '"value": [0-9.]*\|kernel_ms": [0-9.]*' gpurun_out/r3e/ab_$dbg.$rep.json | head -2 | tr '\n' ' ')"
done
done
