#!/bin/bash
# ab_lib.sh LIB [ranks]: the in-tree library against profiles/ubench/libs/LIB (VBNMF_LIB), same box, interleaved:
# it/s by bench.py, then k_update by side from a rocprofv3 kernel trace.
export BENCH_NO_SWEEP=1 BENCH_NO_TRAFFIC=1 TMPDIR=/tmp
R=$PWD
LIB=$R/profiles/ubench/libs/$1
RANKS=${2:-"10 20"}
for rep in 1 2 3; do
  for r in $RANKS; do
    for v in new old; do
      if [ $v = new ]; then unset VBNMF_LIB; else export VBNMF_LIB=$LIB; fi
      out=$(python3 $R/bench.py --rank $r --steps 600 --no-cpu --no-ml 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s step %.1f us sweep %.1f us' % (d['value'], 1e3*d['ms_per_step'], 1e3*d['roofline']['kernel_ms']))")
      echo "rank $r $v rep $rep: $out"
    done
  done
done
cd /tmp
for v in new old; do
  if [ $v = new ]; then unset VBNMF_LIB; else export VBNMF_LIB=$LIB; fi
  rm -rf /tmp/abl_$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/abl_$v -o t -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu --no-ml > /dev/null 2>&1
  echo "rank 10 $v"; python3 $R/profiles/ubench/r04/update_split.py /tmp/abl_$v | grep k_update
done
