#!/bin/bash
# One C5 partition alone on the GPU -- the step a rank of an 8-GPU node runs, minus the exchange partners: step time of the
# shipped library against round 4's first-half build (profiles/ubench/libs/lib_r04_head.so, git 56d43aa), same box,
# interleaved, as a local group of one and through a one-rank RCCL communicator; then the step's time line (rocprofv3
# kernel trace, ubench/r04/step_timeline.py) for the RCCL rank with the pack on the main stream (default) and on the comm stream.
export TMPDIR=/tmp
R=$PWD
ms() { python3 -c "import sys,json; print('%.4f ms per step' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for rep in 1 2; do
  for mode in "" "--rccl"; do
    echo "rep $rep ${mode:-local group of one}: new $(python3 $R/profiles/ubench/r04/c5_one_partition.py $mode 2>/dev/null | ms)   first-half build $(VBNMF_LIB=$R/profiles/ubench/libs/lib_r04_head.so python3 $R/profiles/ubench/r04/c5_one_partition.py $mode 2>/dev/null | ms)"
  done
done
cd /tmp
for v in default 0; do
  rm -rf /tmp/tl
  if [ $v = default ]; then unset VBNMF_PACK_ON_MAIN; else export VBNMF_PACK_ON_MAIN=$v; fi
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 $R/profiles/ubench/r04/c5_one_partition.py --rccl --steps 100 > /dev/null 2>&1
  echo "== one-rank RCCL communicator, VBNMF_PACK_ON_MAIN=$v"; python3 $R/profiles/ubench/r04/step_timeline.py /tmp/tl
done
unset VBNMF_PACK_ON_MAIN
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 $R/profiles/ubench/r04/c5_one_partition.py --steps 100 > /dev/null 2>&1
echo "== local group of one (pack on the comm stream)"; python3 $R/profiles/ubench/r04/step_timeline.py /tmp/tl
