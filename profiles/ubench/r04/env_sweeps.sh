#!/bin/bash
export BENCH_NO_TRAFFIC=1      # no nested rocprofv3 runs under a profiler (bench.py: measure_sweep_traffic)
# env-switch sweeps on the round's final build (same box, run through gpurun): (1) the headline under the sweep's run-time
# knobs now that the cells are ordered, (2) one C5 partition alone under the number of CUs left free for the comm stream.
export BENCH_NO_SWEEP=1
REPO=$PWD
one() {  # label, env assignments...
  label=$1; shift
  out=$(env "$@" python3 $REPO/bench.py --steps 600 --no-cpu --no-ml 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s step %.1f us sweep %.1f us' % (d['value'], 1e3*d['ms_per_step'], 1e3*d['roofline']['kernel_ms']))")
  echo "headline $label: $out"
}
for rep in 1 2; do
  one "default" X=1
  one "PULL_ENDS_A=2" VBNMF_PULL_ENDS_A=2
  one "PULL_ENDS_A=6" VBNMF_PULL_ENDS_A=6
  one "PULL_ENDS_B=2" VBNMF_PULL_ENDS_B=2
  one "MAX_LEN=192" VBNMF_MAX_LEN=192
  one "MAX_LEN=320" VBNMF_MAX_LEN=320
  one "NWG=255" VBNMF_NWG=255
done
for rep in 1 2; do
  for cus in 0 8 16 32; do
    out=$(VBNMF_COMM_CUS=$cus python3 $REPO/profiles/ubench/r04/c5_one_partition.py 2>/dev/null | grep -o 'ms_per_step": [0-9.]*')
    echo "C5 partition alone COMM_CUS=$cus: $out"
  done
done
