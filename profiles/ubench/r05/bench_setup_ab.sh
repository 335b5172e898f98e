#!/bin/bash
# bench_setup_ab.sh "LIBA LIBB": the bench line's own set-up seconds (ingest, engine_create) per library variant, interleaved
export BENCH_NO_TRAFFIC=1 BENCH_NO_SWEEP=1 BENCH_NO_SMALL=1
R=$PWD
for rep in 1 2 3; do
for lib in $1; do
  if [ $lib = tree ]; then unset VBNMF_LIB; else export VBNMF_LIB=$R/profiles/ubench/libs/$lib; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu --no-ml --no-traffic 2>/dev/null > gpurun_out/bsa.log || { echo "[$lib] failed"; continue; }
  python3 - <<PY
import json; d=json.load(open("gpurun_out/bsa.log")); s=d["setup"]; print("[$lib] rep$rep ingest %.3f s engine_create %.3f s set_state %.3f s  value %.0f" % (s["ingest_s"], s["engine_create_s"], s["set_state_s"], d["value"]))
PY
done
done
