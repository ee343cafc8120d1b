#!/bin/bash
# interleaved A/B of library builds on tools/train_bench.py: ab_train_libs.sh <rounds> <lib path | -> ...   ("-" = this tree's product lib)
R=$1; shift
for i in $(seq $R); do for v in "$@"; do
  if [ "$v" = "-" ]; then unset CCN_HIP_LIB; else export CCN_HIP_LIB=$PWD/$v; fi
  echo -n "$v: "; timeout -k 10 300 python tools/train_bench.py --steps 20 --warmup 5 2>&1 | tail -1 || exit 1
done; done
