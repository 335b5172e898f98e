#!/bin/bash
# c5_flags.sh: ONE C5 partition alone on the GPU through a one-rank RCCL communicator (real librccl): the step with the pack inside
# the cell-side sweep and flag hand-overs (default) against the event form with k_pack_tail as a kernel of its own
# (VBNMF_PACK_IN_SWEEP=0); VBNMF_COMM_CUS=0 and 32; then the time lines.
export TMPDIR=/tmp BENCH_NO_TRAFFIC=1
R=$PWD
ms() { python3 -c "import sys,json; print('%.4f ms per step' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for rep in 1 2 3; do
for cus in 0 32; do
for pis in 0 1; do
  echo "rep $rep VBNMF_COMM_CUS=$cus VBNMF_PACK_IN_SWEEP=$pis: $(VBNMF_COMM_CUS=$cus VBNMF_PACK_IN_SWEEP=$pis python3 $R/profiles/ubench/r05/c5_partition.py --rccl 2>/dev/null | ms)"
done
done
done
cd /tmp
for pis in 0 1; do
  rm -rf /tmp/tl
  VBNMF_COMM_CUS=0 VBNMF_PACK_IN_SWEEP=$pis rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 $R/profiles/ubench/r05/c5_partition.py --rccl --steps 100 > /dev/null 2>&1
  echo "== VBNMF_COMM_CUS=0 VBNMF_PACK_IN_SWEEP=$pis"; python3 $R/profiles/ubench/r04/step_timeline.py /tmp/tl
done
