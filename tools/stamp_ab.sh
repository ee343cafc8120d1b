#!/bin/bash
# stamp timelines (diagnostics builds) of the last launch with <tiles>:<ntaps> for several libraries: stamp_ab.sh 256:9 diag ...
L=/root/repo/clip-neural-image-conpression_amd/csrc
export CCN_STAMPS=$1; shift
for v in "$@"; do
  export CCN_HIP_LIB=$L/libccn_hip_$v.so
  timeout -k 10 200 python tools/prof_sample.py --steps 2 > gpurun_out/ps_$v.log 2>&1 || { tail -5 gpurun_out/ps_$v.log; exit 1; }
  cp gpurun_out/stamps.txt gpurun_out/stamps_$v.txt
  echo "== $v"; python tools/stamp_timeline.py gpurun_out/stamps_$v.txt
done
