#!/bin/bash
# same-box A/B of sweep builds: ab.sh libA libB ... ; prints value + sweep kernel ms for each, twice
mkdir -p gpurun_out
for rep in 1 2; do
for lib in "$@"; do
  VBNMF_LIB=$PWD/profiles/ubench/libs/$lib python bench.py --steps 200 --warmup 10 --no-cpu 2>/dev/null > gpurun_out/ab_$lib.$rep.log || exit 1
  echo "$lib rep$rep $(grep -o '"value": [0-9.]*\|kernel_ms": [0-9.]*' gpurun_out/ab_$lib.$rep.log | tr '\n' ' ')"
done
done
