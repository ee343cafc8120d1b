#!/usr/bin/env python3
"""Summarise ccn_internal_dump_stamps_pr output: per block 2 roles x 8 words
(0 start, 1 after prologue, 2 end [100 MHz realtime]; 3 barrier-wait, 4 total, 5 phase a, 6 phase b [shader cycles])."""
import sys, numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.float64).reshape(-1, 2, 8)
t0 = d[:, :, 0].min()
for role, name in ((0, "consumer"), (1, "producer")):
    r = d[:, role]
    tot_us = (r[:, 2] - r[:, 0]) / 100.0
    clk = r[:, 4] / np.maximum(tot_us, 1e-9) / 1e3    # GHz
    print(f"{name}: start {((r[:,0]-t0)/100).mean():7.1f}us  prologue {((r[:,1]-r[:,0])/100).mean():5.1f}us  total {tot_us.mean():7.1f}us (max {tot_us.max():.1f})"
          f"  clock {clk.mean():.2f} GHz  barrier-wait {100*(r[:,3]/r[:,4]).mean():5.1f}%  dump {100*(r[:,5]/r[:,4]).mean():5.1f}%  epilogue {100*(r[:,6]/r[:,4]).mean():5.1f}%  request {100*(r[:,7]/r[:,4]).mean():5.1f}%")
