#!/bin/bash
# abprof.sh libA libB ...: rocprofv3 kernel averages (us) of the bench step for each build, same box
export TMPDIR=/tmp
R=$PWD
for lib in "$@"; do
  rm -rf /tmp/abprof_$lib
  (cd /tmp && VBNMF_LIB=$R/gpurun_build/libs/$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abprof_$lib -o st -- python3 $R/bench.py --steps 300 --warmup 10 --no-cpu --no-ml > /dev/null 2>&1)
  python3 - <<EOF
import csv,glob
f=glob.glob("/tmp/abprof_$lib/**/*kernel_stats.csv",recursive=True)
out=[]
for r in csv.DictReader(open(f[0])):
    n=r["Name"].split("(")[0].replace("void vbnmf::","")
    if n.startswith("k_"): out.append("%s %.2f" % (n.split("<")[0], float(r["AverageNs"])/1e3))
print("$lib", " | ".join(out[:5]))
EOF
done
