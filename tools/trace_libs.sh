#!/bin/bash
# per-launch kernel trace of one UNet forward for several library builds: trace_libs.sh <tag> ...  ("-" = product library)
# (libraries built with `make ab V=_x EXTRA=-D...`; the round-2 timing switches it was used with are gone from the source, docs/EXPERIMENTS.md)
L=/root/repo/clip-neural-image-conpression_amd/csrc
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "-" ]; then unset CCN_HIP_LIB; t=product; else export CCN_HIP_LIB=$L/libccn_hip_$v.so; t=$v; fi
  O=gpurun_out/trace_$t; rm -rf $O; mkdir -p $O
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O -- python3 $OLDPWD/tools/prof_sample.py --steps 3 > /dev/null 2>&1) || exit 1
  python tools/trace_forward.py $O > gpurun_out/forward_$t.txt || exit 1
  tail -1 gpurun_out/forward_$t.txt
done
