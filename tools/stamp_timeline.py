#!/usr/bin/env python3
"""Timeline of one persistent-kernel launch from its in-kernel stamps (diagnostics build, CCN_STAMPS=<tiles>[:<ntaps>]).
Per block 2 roles (consumer wave 0, producer wave 4) x 8 words: 0 start, 1 chunk 0 visible, 2 exit [100 MHz realtime];
3 barrier-wait, 4 total, 5 dump, 6 epilogue, 7 request [shader cycles]."""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.float64).reshape(-1, 2, 8)
d = d[d[:, 0, 0] > 0]
t0 = d[:, :, 0].min()
us = lambda v: (v - t0) / 100.0
c, p = d[:, 0], d[:, 1]
print(f"blocks {len(d)}")
print(f"dispatch skew: first start 0.0, mean {us(c[:,0]).mean():.2f}, last {us(c[:,0]).max():.2f} us")
print(f"consumer: chunk0 visible at {us(c[:,1]).mean():.2f} (from own start {((c[:,1]-c[:,0])/100).mean():.2f}); loop ends {us(c[:,2]).mean():.2f} (max {us(c[:,2]).max():.2f}); "
      f"in-loop {((c[:,2]-c[:,1])/100).mean():.2f} us, of which barrier wait {(c[:,3]/c[:,4]*(c[:,2]-c[:,0])/100).mean():.2f} us")
print(f"producer: exits at {us(p[:,2]).mean():.2f} (max {us(p[:,2]).max():.2f}); tail after consumers {((p[:,2]-c[:,2])/100).mean():.2f} us; "
      f"barrier wait {(p[:,3]/p[:,4]*(p[:,2]-p[:,0])/100).mean():.2f} us, dump {(p[:,5]/p[:,4]*(p[:,2]-p[:,0])/100).mean():.2f}, "
      f"epilogue {(p[:,6]/p[:,4]*(p[:,2]-p[:,0])/100).mean():.2f}, request {(p[:,7]/p[:,4]*(p[:,2]-p[:,0])/100).mean():.2f} us")
print(f"consumer cold start (own clock, us since its start): request done {(c[:,5]/100).mean():.2f}, loads arrived {(c[:,6]/100).mean():.2f}, own rows staged {(c[:,7]/100).mean():.2f}, "
      f"chunk 0 visible {((c[:,1]-c[:,0])/100).mean():.2f}")
clk = p[:, 4] / np.maximum((p[:, 2] - p[:, 0]) / 100.0, 1e-9) / 1e3
print(f"kernel span (first start -> last exit) {us(p[:,2]).max():.2f} us; shader clock {clk.mean():.2f} GHz")
