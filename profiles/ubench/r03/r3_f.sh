#!/bin/bash
mkdir -p gpurun_out/r3f
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3f/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3f/tests.log
for rep in 1 2; do
for nf in 0 1; do
  VBNMF_NO_CONTROL_FOLD=$nf timeout -k 10 300 python bench.py --steps 300 --warmup 10 --no-cpu --no-ml > gpurun_out/r3f/ab_nf$nf.$rep.json 2>gpurun_out/r3f/ab_nf$nf.$rep.err || { echo "bench nf=$nf failed"; tail -3 gpurun_out/r3f/ab_nf$nf.$rep.err; }
  echo "NO_FOLD=$nf rep$rep $(grep -o '"value": [0-9.]*\|kernel_ms": [0-9.]*\|value_host_stepped": [0-9.]*' gpurun_out/r3f/ab_nf$nf.$rep.json | head -3 | tr '\n' ' ') hyper_on $(python3 -c "import json;d=json.load(open('gpurun_out/r3f/ab_nf$nf.$rep.json'));print(round(d['hyper_updates_on']['value'],1))")"
done
done
