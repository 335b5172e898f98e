#!/bin/bash
# lib_ab.sh "LIBA LIBB ..." [ranks] [extra env]: library variants under profiles/ubench/libs (VBNMF_LIB), same box, interleaved,
# three repetitions: it/s of the device-driven loop, host-stepped it/s, k_sweep by the engine's events.
export BENCH_NO_TRAFFIC=1 BENCH_NO_SWEEP=1
R=$PWD
LIBS=$1
RANKS=${2:-"10 20"}
for rank in $RANKS; do
for rep in 1 2 3; do
for lib in $LIBS; do
  if [ $lib = tree ]; then unset VBNMF_LIB; else export VBNMF_LIB=$R/profiles/ubench/libs/$lib; fi
  python3 bench.py --steps 400 --warmup 10 --no-cpu --no-ml --no-traffic --rank $rank 2>/dev/null > gpurun_out/lib_ab.log || { echo "[$lib] failed"; continue; }
  python3 - <<PY
import json; d=json.load(open("gpurun_out/lib_ab.log")); print("rank $rank [$lib] rep$rep value %.1f step %.4f ms host %.1f sweep %.4f ms" % (d["value"], d["ms_per_step"], d["host_stepped"]["value"], d["roofline"]["kernel_ms"]))
PY
done
done
done
