#!/bin/bash
# diagnostics build: sweep of the weight-gradient kernel's workgroup count on the side stream (CCN_WGRAD_WS_WGS) on the training step
export CCN_HIP_LIB=$PWD/clip-neural-image-conpression_amd/csrc/libccn_hip_diag.so
for rep in 1 2; do
for w in ${@:-96 128 160 192 256}; do
  echo -n "wgs=$w: "; CCN_WGRAD_WS_WGS=$w timeout -k 10 200 python tools/train_bench.py --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-90 || exit 1
done; done
