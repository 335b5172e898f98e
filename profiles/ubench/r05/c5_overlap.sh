#!/bin/bash
# c5_overlap.sh: ONE C5 partition alone on the GPU with the all-reduce carried by KERNELS of RCCL's launch
# shape (tests/fake_rccl, FAKE_RCCL_KERNEL=1: FAKE_RCCL_BLOCKS x 512 threads, 4 KB of LDS each, on the comm stream).  Does the
# n x R all-reduce start and finish INSIDE the cell-side sweep?  With how many CUs left free by the sweep (VBNMF_COMM_CUS)?
# What does it take alone?  rocprofv3 kernel trace + profiles/ubench/r04/step_timeline.py.
export TMPDIR=/tmp BENCH_NO_TRAFFIC=1
R=$PWD
export VBNMF_RCCL_LIB=$R/tests/fake_rccl/_build/libfake_rccl.so FAKE_RCCL_KERNEL=1 FAKE_RCCL_TIMEOUT_S=20
ms() { python3 -c "import sys,json; print('%.4f ms per step' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
cd /tmp
echo "== the collective's kernels ALONE (host-stepped steps: nothing else runs beside them), 24 blocks"
rm -rf /tmp/tl; FAKE_RCCL_BLOCKS=24 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 $R/profiles/ubench/r05/c5_partition.py --rccl --host --steps 40 > /dev/null 2>&1
python3 $R/profiles/ubench/r05/kernel_durations.py /tmp/tl k_copy_in k_reduce k_pack
for cus in 8 0 16 32; do
for blocks in 24 64; do
  export VBNMF_COMM_CUS=$cus FAKE_RCCL_BLOCKS=$blocks
  echo "== device-driven loop, VBNMF_COMM_CUS=$cus, collective of $blocks blocks x 512 threads: $(python3 $R/profiles/ubench/r05/c5_partition.py --rccl 2>/dev/null | ms)"
  rm -rf /tmp/tl
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 $R/profiles/ubench/r05/c5_partition.py --rccl --steps 100 > /dev/null 2>&1
  python3 $R/profiles/ubench/r04/step_timeline.py /tmp/tl
done
done
