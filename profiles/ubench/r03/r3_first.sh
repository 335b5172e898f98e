#!/bin/bash
# round 3, first GPU contact: tests, baseline bench, block-width sensitivity A/B, C4 sweep baseline
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q > gpurun_out/r3a/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3a/tests.log
python bench.py --steps 300 --warmup 10 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err || { echo bench failed; tail -5 gpurun_out/r3a/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3a/bench.json'))
print("value",d["value"],"host",d["value_host_stepped"],"sweep_ms",d["roofline"]["kernel_ms"],"setup",d["setup"])
PY
for kb in 160 80; do
  VBNMF_LDS_KB=$kb python bench.py --steps 300 --warmup 10 --no-cpu --no-ml > gpurun_out/r3a/ab_lds$kb.json 2>/dev/null
  echo "LDS_KB=$kb $(grep -o '"value": [0-9.]*\|kernel_ms": [0-9.]*' gpurun_out/r3a/ab_lds$kb.json | head -2 | tr '\n' ' ')"
done
export TMPDIR=/tmp
cd /tmp
for kb in 160 80; do
  VBNMF_LDS_KB=$kb rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3a/stats_lds$kb -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 10 --no-cpu --no-ml > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3a/stats_lds$kb.err
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r3a/stats_lds$kb -name '*kernel_stats.csv' | head -1)
  echo "== LDS_KB=$kb kernel stats"; head -8 $f | cut -c1-160
done
cd $GRAFT_REPO_ROOT
python tests/manual_c4_sweep.py > gpurun_out/r3a/c4.log 2>&1; tail -3 gpurun_out/r3a/c4.log
cp gpurun_out/c4_sweep.json gpurun_out/r3a/c4_sweep_baseline.json
