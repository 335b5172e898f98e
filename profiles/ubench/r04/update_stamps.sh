#!/bin/bash
# in-kernel time line of k_update at ranks 10 and 20 (instrumented library, see update_stamps.py)
export VBNMF_LIB=$PWD/profiles/ubench/libs/lib_r04_STAMPS.so
for r in 10 20; do python3 profiles/ubench/r04/update_stamps.py $r || exit 1; done
