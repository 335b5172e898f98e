#!/bin/bash
export BENCH_NO_TRAFFIC=1      # no nested rocprofv3 runs under a profiler (bench.py: measure_sweep_traffic)
export TMPDIR=/tmp
R=$PWD
cd /tmp
for lib in lib_pack18b.so lib_upd_NOGATHER.so lib_upd_NOSPECIAL.so; do
  export VBNMF_LIB=$R/profiles/ubench/libs/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$lib -o s -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu --no-ml > /dev/null 2>&1
  f=$(find $R/gpurun_out/ab_$lib -name "*kernel_stats.csv" | head -n 1)
  echo "== $lib"; cut -d, -f1-4 $f | sed -E 's/\(.*\)"/"/' | grep -E "k_update|k_sweep|k_final|k_control" | cut -c1-90
done
