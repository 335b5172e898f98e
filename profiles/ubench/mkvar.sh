#!/bin/bash
# mkvar.sh NAME [extra hipcc flags]: compile the working tree into gpurun_build/libs/lib_NAME.so
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -shared "$@" -o gpurun_build/libs/lib_$name.so -x hip ccfindr_amd/csrc/host.cpp ccfindr_amd/csrc/mtx.cpp ccfindr_amd/csrc/engine.hip -pthread 2>&1 | grep -E "error|spill" 
ls -la gpurun_build/libs/lib_$name.so
