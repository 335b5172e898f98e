#!/bin/bash
mkdir -p gpurun_out/r3b
python -m pytest tests -m gpu -x -q > gpurun_out/r3b/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3b/tests.log
python bench.py --steps 300 --warmup 10 --no-cpu --no-ml > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err || { echo bench failed; tail -5 gpurun_out/r3b/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3b/bench.json'))
print("value",d["value"],"host",d["value_host_stepped"],"sweep_ms",d["roofline"]["kernel_ms"],"setup",{k:v for k,v in d["setup"].items() if k!="note"})
PY
VBNMF_BUILD_TIMES=1 python tests/manual_c4_sweep.py > gpurun_out/r3b/c4.log 2>&1; grep -v "layout side" gpurun_out/r3b/c4.log | tail -24; grep "layout side" gpurun_out/r3b/c4.log | head -20
cp gpurun_out/c4_sweep.json gpurun_out/r3b/c4_sweep_classes1.json
python tests/manual_c4_sweep.py --classes 2 > gpurun_out/r3b/c4_2.log 2>&1; tail -2 gpurun_out/r3b/c4_2.log
cp gpurun_out/c4_sweep.json gpurun_out/r3b/c4_sweep_classes2.json
