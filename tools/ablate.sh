#!/bin/bash
# usage: tools/ablate.sh "0 1 2 4 8 3 7 15"  -- per-kernel times of the ws kernel under each CCN_DBG ablation mask
export TMPDIR=/tmp
for m in $1; do
  rm -rf gpurun_out/abl_$m
  CCN_DBG=$m timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abl_$m -- python3 tools/prof_sample.py --steps 2 > gpurun_out/abl_$m.log 2>&1
  echo "== CCN_DBG=$m"; python3 tools/trace_summary.py gpurun_out/abl_$m | grep -E "conv_ws" 
done
