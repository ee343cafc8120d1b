#!/usr/bin/env python3
"""Summarise gpurun_out/stamps.txt: per role (consumer, B producer, A producer) phase durations in microseconds.
s_memrealtime ticks at 100 MHz (10 ns)."""
import sys, numpy as np
a = np.loadtxt(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stamps.txt", dtype=np.uint64).astype(np.float64).reshape(-1, 3, 8)
wait = a[:, :, 4].copy(); clk = (a[:, :, 6] - a[:, :, 5]) / np.maximum((a[:, :, 2] - a[:, :, 1]) * 0.01, 1e-9) / 1e3   # GHz
t0 = a[:, :, 0].min()
a = (a - t0) * 0.01   # us
names = ["consumer", "Bprod", "Aprod"]
print(f"blocks {a.shape[0]}  kernel span {a[:, :, 3].max():.1f} us")
for r in range(3):
    pro = a[:, r, 1] - a[:, r, 0]; loop = a[:, r, 2] - a[:, r, 1]; epi = a[:, r, 3] - a[:, r, 2]
    print(f"{names[r]:9s} barrier-wait cycles in loop: median {np.median(wait[:, r]):9.0f}   in-loop shader clock {np.median(clk[:, r]):.2f} GHz")
    print(f"{names[r]:9s} prologue {np.median(pro):6.2f} (p90 {np.percentile(pro,90):6.2f})  loop {np.median(loop):6.2f} (p90 {np.percentile(loop,90):6.2f})  epilogue {np.median(epi):6.2f} (p90 {np.percentile(epi,90):6.2f}) us")
start = np.sort(a[:, 0, 0]); end = np.sort(a[:, 0, 3])
print("block start times (us) quantiles:", np.percentile(start, [0, 10, 50, 90, 100]).round(1))
print("block end   times (us) quantiles:", np.percentile(end, [0, 10, 50, 90, 100]).round(1))
