#!/bin/bash
# final collection, part 1: tests + the full profile set of the headline
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_tests.log
bash profiles/collect.sh r03 2>&1 | tail -12
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_k20.json 2>/dev/null
python3 - <<'PY'
import json
for f in ("gpurun_out/r03_bench.json","gpurun_out/r03_bench_k20.json"):
    d=json.load(open(f)); print(f, "value", round(d["value"],1), "host", round(d["value_host_stepped"],1), "frac", round(d["roofline"]["frac"],4), "sweep_ms", round(d["roofline"]["kernel_ms"],5), "cpu", round(d["cpu_baseline"]["value"],2), "elbo_err", d.get("elbo_rel_err_first_steps"))
PY
