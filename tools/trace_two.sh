#!/bin/bash
# launch-by-launch kernel trace of one UNet forward (last DDIM step of tools/prof_sample.py) for this tree and for older worktrees, one box:
#   bash tools/trace_two.sh <outdir> [.cmp_r02 ...]
O=$1; shift; mkdir -p $O; export TMPDIR=/tmp
R=$PWD
for d in . "$@"; do
  n=$(echo $d | tr -d './'); n=${n:-head}
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$n -- python3 $R/$d/tools/prof_sample.py --steps 3 > /dev/null 2>&1)
  python tools/trace_forward.py $O/trace_$n > $O/forward_$n.txt
  rm -rf $O/trace_$n
done
