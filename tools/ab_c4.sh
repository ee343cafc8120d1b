#!/bin/bash
# A/B of library builds on the C4 workload (512 px, base 192, (1,2,2,4), 100 steps, batch 4): ab_c4.sh rounds lib ...  ("-" = product)
R=$1; shift
L=/root/repo/clip-neural-image-conpression_amd/csrc
for i in $(seq $R); do for v in "$@"; do
  if [ "$v" = "-" ]; then unset CCN_HIP_LIB; else export CCN_HIP_LIB=$L/libccn_hip_$v.so; fi
  timeout -k 10 400 python bench.py --size 512 --base 192 --ch-mult 1,2,2,4 --ddim-steps 100 --batch 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'])" || exit 1
done; done
