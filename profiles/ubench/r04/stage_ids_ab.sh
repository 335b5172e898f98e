#!/bin/bash
export BENCH_NO_TRAFFIC=1      # no nested rocprofv3 runs under a profiler (bench.py: measure_sweep_traffic)
# k_update with the inverse index staged in LDS (default) against the global-read form (VBNMF_NO_STAGE_IDS=1): same box,
# interleaved, ranks 10 and 20; it/s by bench.py, per-kernel times by rocprofv3 for one repetition.
export BENCH_NO_SWEEP=1 TMPDIR=/tmp
REPO=$PWD
for rep in 1 2; do
  for r in 10 20; do
    for v in 0 1; do
      out=$(VBNMF_NO_STAGE_IDS=$v python3 $REPO/bench.py --rank $r --steps 600 --no-cpu --no-ml 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s step %.1f us host-stepped %.1f sweep %.1f us' % (d['value'], 1e3*d['ms_per_step'], d['value_host_stepped'], 1e3*d['roofline']['kernel_ms']))")
      echo "rank $r NO_STAGE_IDS=$v rep $rep: $out"
    done
  done
done
cd /tmp
for v in 0 1; do
  rm -rf /tmp/si_$v
  VBNMF_NO_STAGE_IDS=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/si_$v -o t -- python3 $REPO/bench.py --steps 100 --warmup 5 --no-cpu --no-ml > /dev/null 2>&1
  echo "rank 10 NO_STAGE_IDS=$v"; python3 $REPO/profiles/ubench/r04/update_split.py /tmp/si_$v
done
