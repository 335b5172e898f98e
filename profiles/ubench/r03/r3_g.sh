#!/bin/bash
for ml in default 48 64 96 128 256; do
  if [ $ml = default ]; then unset VBNMF_MAX_LEN; else export VBNMF_MAX_LEN=$ml; fi
  timeout -k 10 120 python gpurun_build/small_len.py c2 2>/dev/null
done
unset VBNMF_MAX_LEN
VBNMF_NO_CONTROL_FOLD=1 timeout -k 10 120 python gpurun_build/small_len.py c2 2>/dev/null
for ml in default 8 16 32 64; do
  if [ $ml = default ]; then unset VBNMF_MAX_LEN; else export VBNMF_MAX_LEN=$ml; fi
  timeout -k 10 120 python gpurun_build/small_len.py c1 2>/dev/null
done
unset VBNMF_MAX_LEN
VBNMF_NO_CONTROL_FOLD=1 timeout -k 10 120 python gpurun_build/small_len.py c1 2>/dev/null
