#!/bin/bash
# stamps of the free-running kernel's 2048-block launch under each CCN_DBG mask
export CCN_CONV_DMA=2
for m in $1; do
  CCN_DBG=$m CCN_STAMPS=${2:-2048} timeout -k 10 120 python3 tools/prof_sample.py --steps 1 > gpurun_out/st.log 2>&1
  echo "== CCN_DBG=$m"; python3 tools/stamp_summary.py | sed -n 2,3p
done
