#!/bin/bash
# usage: ab.sh rounds lib1 lib2 ...   ("-" = product lib)
R=$1; shift
L=/root/repo/clip-neural-image-conpression_amd/csrc
for i in $(seq $R); do for v in "$@"; do
  if [ "$v" = "-" ]; then unset CCN_HIP_LIB; else export CCN_HIP_LIB=$L/libccn_hip_$v.so; fi
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'])" || exit 1
done; done
