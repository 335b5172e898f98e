#!/bin/bash
export BENCH_NO_TRAFFIC=1      # no nested rocprofv3 runs under a profiler (bench.py: measure_sweep_traffic)
# k_update taken apart by compile-time ablation (-DVBNMF_ABL_NOGATHER / _NOSPECIAL / _NOWRITE; results are garbage, the times
# are what is read): per-side mean duration by rocprofv3 kernel trace at ranks 10 and 20, same box.
export TMPDIR=/tmp BENCH_NO_SWEEP=1
R=$PWD
cd /tmp
for r in 10 20; do
  for lib in default lib_r04_NOGATHER.so lib_r04_NOSPECIAL.so lib_r04_NOWRITE.so; do
    if [ $lib = default ]; then unset VBNMF_LIB; else export VBNMF_LIB=$R/profiles/ubench/libs/$lib; fi
    rm -rf /tmp/abl
    rocprofv3 --kernel-trace --output-format csv -d /tmp/abl -o t -- python3 $R/bench.py --rank $r --steps 100 --warmup 5 --no-cpu --no-ml > /dev/null 2>&1
    echo "== rank $r $lib"; python3 $R/profiles/ubench/r04/update_split.py /tmp/abl | grep k_update
  done
done
