#!/bin/bash
# A/B of library builds on the training step (bench_train.py): ab_train.sh rounds lib ...  ("-" = product)
R=$1; shift
L=/root/repo/clip-neural-image-conpression_amd/csrc
for i in $(seq $R); do for v in "$@"; do
  if [ "$v" = "-" ]; then unset CCN_HIP_LIB; else export CCN_HIP_LIB=$L/libccn_hip_$v.so; fi
  timeout -k 10 300 python bench_train.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])" || exit 1
done; done
